#!/bin/bash
# Kernel timeline of a short bench run (the driver's K = 20, W = 5): tools/trace_bench.sh [bench args]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/bench_trace
rm -rf $OUT
timeout 240 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 "$@" > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-160
python - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/bench_trace"
ev = []
for f in glob.glob(root + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "traverseKernel" in n or "chainPublish" in n or "envShade" in n:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Queue_Id", 0) or 0), int(r.get("Grid_Size", 0) or 0), "T" if "traverse" in n else ("e" if "envShade" in n else "p")))
ev.sort()
ev = ev[-76:]          # warm-up (5 + the bench's own first calls) + 20 timed (+ their publish kernels)
t0 = ev[0][0]
for s, e, q, g, k in ev:
    print("  %s start %8.1f end %8.1f dur %7.1f us  queue %d grid %d" % (k, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, g))
PY
