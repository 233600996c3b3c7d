cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3_gputest5.log; tail -4 gpurun_out/r3_gputest5.log
timeout 200 python tools/gpu_gather2.py 2>&1 | grep "two ranks"
timeout 600 python tools/gpu_wide.py 43 50 '{"kernel_variant":50,"coop_same_pct":100}' 2>&1 | tee gpurun_out/r3_v10_wide5.log
b() { python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
echo "== V8"; for k in 1 2; do b --steps 200 --warmup 20; b --steps 20 --warmup 5; done
echo "== V10 (variant 50)"; for k in 1 2; do b --steps 200 --warmup 20 --engine-opts '{"kernel_variant":50}'; b --steps 20 --warmup 5 --engine-opts '{"kernel_variant":50}'; done
for v in 43 50; do RACC_BENCH_ARGS="--engine-opts {\"kernel_variant\":$v}" tools/pmc_probe.sh v$v "WRITE_SIZE" "FETCH_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "TD_TD_BUSY_sum TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE" 2>&1 | tail -12; done
