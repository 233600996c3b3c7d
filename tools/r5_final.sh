#!/bin/bash
# round 5, closing GPU run: the suite as the driver runs it, the smoke entry, the bench lines (driver's K = 20 and K = 200)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r5_final_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r5_final_pytest.log; tail -4 gpurun_out/r5_final_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05_k20.json 2> gpurun_out/bench_r05_k20.err; echo "bench rc=$?"
timeout 900 python bench.py > gpurun_out/bench_r05_default.json 2> gpurun_out/bench_r05_default.err; echo "bench default rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --quality 0 --no-extras > gpurun_out/bench_r05_k20_q0.json 2> gpurun_out/bench_r05_k20_q0.err
python - <<'PY'
import json
for f in ("bench_r05_k20", "bench_r05_default", "bench_r05_k20_q0"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], "iso", r.get("kernel_ms_avg"), "frac", r.get("frac"), "stale", r.get("profile_stale"),
              (d.get("reference_builder_tree") or {}).get("mrays_per_s_same_loop_as_value"), d.get("batch_scaling"), (d.get("one_launch_at_a_time") or {}), (d.get("coherent_1M") or {}))
    except Exception as e:
        print(f, "failed", e)
PY
# how often does the suite's chained-launch test exercise the lazy chain's catch-up path (a batch the chain did not reach)?
RACC_CHAIN_LOG=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_chained_launches" -s 2>&1 | grep -c "catch-up kernel" | sed 's/^/catch-up kernels in test_chained_launches*: /'
