#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/probe6
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python tools/gpu_chain_stress.py 25 5 > "$OUT/stress.txt" 2> "$OUT/stress.err"
RACC_NODE_ORDER=0 timeout 900 python tools/gpu_chain_stress.py 15 6 > "$OUT/stress_order0.txt" 2>> "$OUT/stress.err"
RACC_HOSTPIPE_SAME=1 RACC_HOSTPIPE_LANES=4 timeout 600 python tools/gpu_hostpipe.py 8 1048576 > "$OUT/hostpipe.txt" 2> "$OUT/hostpipe.err"
cat "$OUT/stress.txt" "$OUT/stress_order0.txt" "$OUT/hostpipe.txt"; tail -n 5 "$OUT/stress.err" "$OUT/hostpipe.err"
