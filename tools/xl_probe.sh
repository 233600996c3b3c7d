#!/bin/bash
# Round-4 probe: the XL scenes (tools/gpu_xl.py) timed, then FETCH/WRITE/L2-hit PMC passes of the same command; the PCIe microbenchmark.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/xl_probe
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
tools/microbench/pcie > "$OUT/pcie.txt" 2>&1
for g in 2400 3400; do for k in diffuse random; do
  timeout 600 python tools/gpu_xl.py $g $k 0 12 >> "$OUT/xl.txt" 2>> "$OUT/xl.err"
done; done
timeout 600 python tools/gpu_xl.py 3400 diffuse 50 12 >> "$OUT/xl.txt" 2>> "$OUT/xl.err"
cd /tmp && export TMPDIR=/tmp
for k in diffuse random; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TD_TD_BUSY_sum TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 5 400 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_${k}_$i" -- python $REPO/tools/gpu_xl.py 3400 $k 0 6 > "$OUT/pmc_${k}_$i.log" 2>&1 || echo "pass $k $i failed"
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
cat "$OUT/xl.txt"; cat "$OUT/pcie.txt"; tail -5 "$OUT/xl.err"
