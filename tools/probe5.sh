#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/probe5
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest.log"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?" >> "$OUT/bench.err"
tail -n 12 "$OUT/pytest.log"; tail -n 5 "$OUT/bench.err"; python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "pre_timed_launches", "host_buffers_page_locked", "host_buffers_pcie_inclusive_mrays_per_s", "one_launch_at_a_time", "coherent_1M", "batch_scaling", "path_tracer_1080p") if k in d}, indent=1))
print(json.dumps(d["roofline"], indent=1)[:6000])
print(json.dumps(d.get("roofline_by_config"), indent=1))
PY
