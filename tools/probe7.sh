#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/probe7
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
fails=0
for i in $(seq 1 12); do
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "chained or lane_auto or concurrent_lanes or reuse or many_streams" > "$OUT/run_$i.log" 2>&1 || { fails=$((fails+1)); cp "$OUT/run_$i.log" "$OUT/FAILED_$i.log"; }
done
echo "chained subset: $fails failures of 12" | tee "$OUT/summary.txt"
# the whole suite twice more, in file order (the failure of probe5 came after 38 other tests in the same process)
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q > "$OUT/suite_$i.log" 2>&1; echo "suite $i rc $?" | tee -a "$OUT/summary.txt"
  tail -n 4 "$OUT/suite_$i.log" >> "$OUT/summary.txt"
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?" >> "$OUT/summary.txt"
python - "$OUT/bench.json" >> "$OUT/summary.txt" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "host_buffers_page_locked", "one_launch_at_a_time") if k in d}))
PY
cat "$OUT/summary.txt"; grep -h "FAILED\|Error" "$OUT"/suite_*.log | head -20
