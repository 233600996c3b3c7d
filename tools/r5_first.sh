#!/bin/bash
# round 5, first GPU session: the suite (quality-tree tests included), the bench line on both trees
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_pytest.log
tail -5 gpurun_out/r5_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05_k20.json 2> gpurun_out/bench_r05_k20.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --quality 0 --no-extras > gpurun_out/bench_r05_k20_q0.json 2> gpurun_out/bench_r05_k20_q0.err
timeout 600 python bench.py --steps 200 --warmup 20 --no-extras > gpurun_out/bench_r05_k200.json 2> gpurun_out/bench_r05_k200.err
timeout 600 python bench.py --steps 200 --warmup 20 --no-extras --quality 0 > gpurun_out/bench_r05_k200_q0.json 2> gpurun_out/bench_r05_k200_q0.err
python - <<'PY'
import json
for f in ("bench_r05_k20", "bench_r05_k20_q0", "bench_r05_k200", "bench_r05_k200_q0"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], "iso", r.get("kernel_ms_avg"), "frac", r.get("frac"), {k: (v.get("mrays_per_s_same_loop_as_value") if isinstance(v, dict) else None) for k, v in d.items() if k in ("reference_builder_tree", "compressed_wide_kernel_variant_50")}, d.get("batch_scaling"))
    except Exception as e:
        print(f, "failed", e)
PY
