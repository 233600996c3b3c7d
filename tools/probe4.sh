#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/probe4
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
echo "== default" > "$OUT/hostpipe.txt"
timeout 600 python tools/gpu_hostpipe.py 16 1048576 >> "$OUT/hostpipe.txt" 2> "$OUT/hostpipe.err"
echo "== D2H on lane" >> "$OUT/hostpipe.txt"
RACC_HOST_D2H=lane timeout 600 python tools/gpu_hostpipe.py 16 1048576 >> "$OUT/hostpipe.txt" 2>> "$OUT/hostpipe.err"
echo "== 8 hw queues" >> "$OUT/hostpipe.txt"
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/gpu_hostpipe.py 16 1048576 >> "$OUT/hostpipe.txt" 2>> "$OUT/hostpipe.err"
echo "== 8 hw queues, D2H on lane" >> "$OUT/hostpipe.txt"
GPU_MAX_HW_QUEUES=8 RACC_HOST_D2H=lane timeout 600 python tools/gpu_hostpipe.py 16 1048576 >> "$OUT/hostpipe.txt" 2>> "$OUT/hostpipe.err"
cd /tmp && export TMPDIR=/tmp
RACC_HOSTPIPE_LANES=4 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- python $REPO/tools/gpu_hostpipe.py 8 1048576 > "$OUT/trace.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K:" + r["Kernel_Name"][:20]))
for f in glob.glob(out + "/trace/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C:" + r["Direction"][:24]))
ev.sort()
tail = ev[-120:]
start = 0
for i in range(1, len(tail)):
    if tail[i][0] - max(e[1] for e in tail[:i]) > 400000: start = i
grp = tail[start:]
t0 = grp[0][0]
with open(out + "/timeline.txt", "w") as f:
    f.write("events %d span_us %.1f\n" % (len(grp), (max(e[1] for e in grp) - t0) / 1e3))
    for s, e, n in grp:
        f.write("  %8.1f %8.1f %7.1f %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
cat "$OUT/hostpipe.txt"; head -70 "$OUT/timeline.txt"; tail -n 3 "$OUT/hostpipe.err" "$OUT/trace.log"
