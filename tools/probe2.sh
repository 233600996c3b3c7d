#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/probe2
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 600 python tools/gpu_hostpipe.py 8 1048576 > "$OUT/hostpipe.txt" 2> "$OUT/hostpipe.err"
timeout 600 python tools/gpu_hostpipe.py 16 262144 >> "$OUT/hostpipe.txt" 2>> "$OUT/hostpipe.err"
for g in 700 3400; do timeout 600 python tools/gpu_xl.py $g aerial 0 12 >> "$OUT/xl.txt" 2>> "$OUT/xl.err"; done
timeout 600 python tools/gpu_xl.py 3400 aerial 50 12 >> "$OUT/xl.txt" 2>> "$OUT/xl.err"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TD_TD_BUSY_sum TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 400 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_aerial_$i" -- python $REPO/tools/gpu_xl.py 3400 aerial 0 6 > "$OUT/pmc_aerial_$i.log" 2>&1 || echo "pass $i failed"
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
cat "$OUT/xl.txt" "$OUT/hostpipe.txt"; tail -5 "$OUT/xl.err" "$OUT/hostpipe.err"
