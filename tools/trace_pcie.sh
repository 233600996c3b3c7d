cd /tmp && export TMPDIR=/tmp
for sl in 262144 131072; do
  RACC_SLICE=$sl timeout 240 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pcie_trace_$sl -- python $GRAFT_REPO_ROOT/tools/gpu_pcie.py > $GRAFT_REPO_ROOT/gpurun_out/pcie_trace_$sl.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/pcie_trace_$sl.log
done
python - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for sl in (262144, 131072):
    ev = []
    for f in glob.glob("%s/pcie_trace_%d/*/*_kernel_trace.csv" % (root, sl)):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K:" + r["Kernel_Name"][:24]))
    for f in glob.glob("%s/pcie_trace_%d/*/*_memory_copy_trace.csv" % (root, sl)):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C:" + r["Direction"][:20]))
    ev.sort()
    # last call: take the last 60 events, find the last group (gap > 300 us before it)
    tail = ev[-80:]
    start = 0
    for i in range(1, len(tail)):
        if tail[i][0] - max(e[1] for e in tail[:i]) > 150000: start = i
    grp = tail[start:]
    t0 = grp[0][0]
    print("slice", sl, "events", len(grp), "span_us", (max(e[1] for e in grp) - t0) / 1e3)
    for s, e, n in grp:
        print("  %8.1f %8.1f %7.1f %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
