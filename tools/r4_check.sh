#!/bin/bash
# Round 4 closing check: the GPU suite twice, smoke, the default bench line with its wall time.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for i in 1 2; do python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_r04_$i.log 2>&1; echo "suite $i rc $?"; grep "passed\|failed" gpurun_out/pytest_gpu_r04_$i.log | tail -1; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r04_k20.json 2> gpurun_out/bench_r04_k20.err; echo "bench wall $SECONDS s"
python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > gpurun_out/bench_r04_k200.json 2>/dev/null
python - <<'PY'
import json
for f in ("gpurun_out/bench_r04_k20.json", "gpurun_out/bench_r04_k200.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
PY
