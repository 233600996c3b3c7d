#!/bin/bash
# Round 4's closing GPU run: suite, microbenchmarks, rocprofv3 passes (tools/profile_bench.sh r04), timeline, bench lines.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu_r04.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu_r04.log"
tail -n 6 "$OUT/pytest_gpu_r04.log"
[ -x tools/microbench/pcie ] || (cd tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o pcie pcie.hip)
bash tools/microbench/run_microbench.sh r04 > "$OUT/microbench_r04.log" 2>&1
tools/microbench/pcie > "$OUT/microbench_r04/pcie.txt" 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_r04_k20.json" 2> "$OUT/bench_r04_k20.err"; echo "bench k20 rc $?"
timeout 900 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > "$OUT/bench_r04_k200.json" 2> "$OUT/bench_r04_k200.err"; echo "bench k200 rc $?"
bash tools/profile_bench.sh r04 > "$OUT/profile_r04.log" 2>&1
tail -n 4 "$OUT/profile_r04.log"
bash tools/trace_bench.sh > "$OUT/bench_trace_r04.txt" 2>&1
python - "$OUT/bench_r04_k20.json" "$OUT/bench_r04_k200.json" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], json.dumps({k: d[k] for k in ("value", "ms_per_step", "host_buffers_page_locked", "one_launch_at_a_time", "coherent_1M", "batch_scaling") if k in d}))
    except Exception as e:
        print(f, "unreadable", e)
PY
