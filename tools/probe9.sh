#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/probe9
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
RACC_HOSTPIPE_DEBUG=1 RACC_HOSTPIPE_LANES=4 timeout 600 python tools/gpu_hostpipe.py 8 1048576 > "$OUT/hostpipe.txt" 2> "$OUT/hostpipe.err"
cat "$OUT/hostpipe.txt"; tail -n 3 "$OUT/hostpipe.err"
bash tools/probe8.sh
