#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/probe8
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python tools/gpu_policy_sweep.py '{}' '{"tail_active":16}' '{"tail_active":24}' '{"tail_active":48}' '{"tail_active":64}' \
  '{"thin_reps":4}' '{"thin_reps":16}' '{"thin_reps":32}' '{"drain_prefetch":1}' '{"drain_prefetch":1,"tail_active":48}' \
  '{"tail_active":48,"thin_reps":16}' '{"tail_active":64,"thin_reps":16}' '{"refill_min":8}' '{"refill_min":16}' '{"leaf_min":6}' '{"leaf_min":14}' \
  '{"inner_reps":2}' '{"inner_reps":4}' '{"coop_same_pct":10}' '{"coop_same_pct":40}' '{"chunk":128}' '{}' > "$OUT/sweep.txt" 2> "$OUT/sweep.err"
cat "$OUT/sweep.txt"; tail -n 3 "$OUT/sweep.err"
