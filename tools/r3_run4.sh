cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3_gputest4.log; tail -4 gpurun_out/r3_gputest4.log
timeout 600 tools/microbench/run_microbench.sh r03 2>&1 | tail -20
timeout 200 python tools/gpu_gather2.py 2>&1 | tail -3
echo "== trace default"; timeout 300 tools/trace_bench.sh > gpurun_out/r3_trace_v8.log 2>&1; tail -60 gpurun_out/r3_trace_v8.log
b() { python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for f in 0 2; do echo "== RACC_DEBUG_FLAGS=$f (bit1 = write-through results)"; for k in 1 2; do RACC_DEBUG_FLAGS=$f b --steps 200 --warmup 20; RACC_DEBUG_FLAGS=$f b --steps 20 --warmup 5; done; done
echo "== V10 (variant 50)"; for k in 1 2; do b --steps 200 --warmup 20 --engine-opts '{"kernel_variant":50}'; b --steps 20 --warmup 5 --engine-opts '{"kernel_variant":50}'; done
echo "== V10 write-through"; RACC_DEBUG_FLAGS=2 b --steps 200 --warmup 20 --engine-opts '{"kernel_variant":50}'; RACC_DEBUG_FLAGS=2 b --steps 20 --warmup 5 --engine-opts '{"kernel_variant":50}'
for f in 0 2; do RACC_DEBUG_FLAGS=$f tools/pmc_probe.sh wt$f "WRITE_SIZE" "FETCH_SIZE" 2>&1 | tail -5; done
