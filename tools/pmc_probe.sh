#!/bin/bash
# Ad-hoc PMC passes over bench.py's timed launches (run on the GPU box): tools/pmc_probe.sh <tag> "<counters of pass 1>" "<pass 2>" ...
# Each pass is its own rocprofv3 run (--pmc only, never combined with traces).  Prints the mean per traversal launch.
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 ${RACC_BENCH_ARGS}"
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 5 180 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- $CMD > "$OUT/p$i.log" 2>&1 || tail -3 "$OUT/p$i.log"
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = {}
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    byc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "traverseKernel" in r["Kernel_Name"]:
            byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in byc.items():
        v.sort(); vals = [x[1] for x in v]
        out[k] = round(sum(vals[4:24]) / max(1, len(vals[4:24])), 1)
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
