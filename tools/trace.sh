#!/bin/bash
# rocprofv3 timelines (on the GPU box):   tools/trace.sh bench [bench args]  |  tools/trace.sh render
MODE=${1:-bench}; shift
if [ "$MODE" = bench ]; then
# Kernel timeline of a short bench run (the driver's K = 20, W = 5): tools/trace.sh bench [bench args]
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
else
# Kernel + memory-copy timeline of racc::render with callbacks that cost nothing (render_check --null-callbacks): where does the host RayStream
# path under the scheduler lose against the link's rate?  tools/trace.sh render   (on the GPU box; prints a digest + the last frame's events)
cd "${GRAFT_REPO_ROOT:-.}"; export GRAFT_REPO_ROOT=$(pwd)
python -u - <<'PY'
import sys
sys.path.insert(0, ".")
from rayaccel_amd import synth
synth.write_scene_bin("/tmp/scene1080.bin", synth.battlefield_synth(), viewport=(1920, 1080))
PY
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/render_trace; rm -rf $OUT
RACC_CPU_THREADS=16 timeout 240 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- $GRAFT_REPO_ROOT/tests/cpp/render_check /tmp/scene1080.bin --null-callbacks 1920 1080 16 3 > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-300
python - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/render_trace"
ev = []
for f in glob.glob(root + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + ("traverse" if "traverse" in r["Kernel_Name"] else ("envShade" if "envShade" in r["Kernel_Name"] else r["Kernel_Name"][:20])), 0))
for f in glob.glob(root + "/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r["Direction"][-14:], 0))
ev.sort()
# the last frame: events after the last gap of > 2 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 2_000_000: cut = i
fr = ev[cut:]
t0, t1 = fr[0][0], max(e[1] for e in fr)
print("last frame: %d events over %.2f ms" % (len(fr), (t1 - t0) / 1e6))
for kind in ("C HOST_TO_DEVICE", "C DEVICE_TO_HOST", "K traverse", "K envShade"):
    sel = [e for e in fr if e[2].startswith(kind[:2]) and kind[2:] in e[2]]
    if not sel: continue
    busy = sum(e[1] - e[0] for e in sel)
    # union of intervals (kernels overlap)
    u, cur_s, cur_e = 0, None, None
    for s, e, _, _ in sorted(sel):
        if cur_e is None or s > cur_e:
            if cur_e is not None: u += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    u += cur_e - cur_s
    durs = sorted((e[1] - e[0]) / 1e3 for e in sel)
    print("%-18s n %4d  sum %.2f ms  union %.2f ms (%.0f %% of the frame)  median %.1f us  p90 %.1f us" % (kind, len(sel), busy / 1e6, u / 1e6, 100.0 * u / (t1 - t0), durs[len(durs) // 2], durs[int(len(durs) * 0.9)]))
for s, e, n, _ in fr[:70]:
    print("  %9.1f %9.1f %8.1f %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
fi
