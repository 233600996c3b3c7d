#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/probe3
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest.log"
timeout 600 python tools/gpu_hostpipe.py 8 1048576 > "$OUT/hostpipe.txt" 2> "$OUT/hostpipe.err"
timeout 600 python tools/gpu_hostpipe.py 16 262144 >> "$OUT/hostpipe.txt" 2>> "$OUT/hostpipe.err"
timeout 900 python tools/gpu_order_ab.py 3400 0,50 > "$OUT/order.txt" 2> "$OUT/order.err"
cd /tmp && export TMPDIR=/tmp
for o in 0 1; do i=0
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    RACC_NODE_ORDER=$o timeout -k 5 400 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_random${o}_$i" -- python $REPO/tools/gpu_xl.py 3400 random 0 6 > "$OUT/pmc_random${o}_$i.log" 2>&1 || echo "pass $o $i failed"
  done
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/pmc_*_[0-9]")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        byc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "traverseKernel" in r["Kernel_Name"]:
                byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in byc.items():
            v.sort(); vals = [x[1] for x in v][-4:]
            res.setdefault(d.split("/")[-1].rsplit("_", 1)[0], {})[c] = sum(vals) / len(vals)
json.dump(res, open(out + "/pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
tail -n 15 "$OUT/pytest.log"; cat "$OUT/hostpipe.txt" "$OUT/order.txt"; tail -n 5 "$OUT/hostpipe.err" "$OUT/order.err"
