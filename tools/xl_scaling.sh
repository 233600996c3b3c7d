#!/bin/bash
# battlefield-synth at four sizes, 1M incoherent rays: kernel time, algorithmic bytes, fabric traffic and L2 hit rate (separate --pmc passes):
# where the kernel leaves the caches.  Writes gpurun_out/xl_scaling/summary.json (copied to profiles/<round>/xl_scaling.json).
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/xl_scaling
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for g in 700 1400 2400 3400; do
  python $REPO/tools/gpu_xl.py $g random 0 12 > "$OUT/time_$g.json" 2>/dev/null
  timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$g" -- python $REPO/tools/gpu_xl.py $g random 0 6 > /dev/null 2>&1
  timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_$g" -- python $REPO/tools/gpu_xl.py $g random 0 6 > /dev/null 2>&1
  timeout -k 5 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/hit_$g" -- python $REPO/tools/gpu_xl.py $g random 0 6 > /dev/null 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = []
for g in (700, 1400, 2400, 3400):
    t = json.loads(open("%s/time_%d.json" % (out, g)).read().strip().splitlines()[-1])
    c = {}
    for kind in ("fetch", "write", "hit"):
        for f in glob.glob("%s/%s_%d/*/*_counter_collection.csv" % (out, kind, g)):
            byc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "traverseKernel" in r["Kernel_Name"]:
                    byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            for k, v in byc.items():
                v.sort(); vals = [x[1] for x in v][-4:]; c[k] = sum(vals) / len(vals)
    traffic = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
    rows.append(dict(grid=g, triangles=t["triangles"], device_mb=round(t["node_mb"] + t["pair_mb"], 1), kernel_ms=t["ms"], mrays_per_s=t["mrays"], nv=t["nv"], np=t["np"],
                     algorithmic_bytes=t["alg_bytes"], frac_of_hbm_peak=t["alg_frac_hbm"], fabric_bytes=int(traffic), traffic_frac_of_algorithmic=round(traffic / t["alg_bytes"], 3),
                     fabric_frac_of_hbm_peak=round(traffic / (t["ms"] * 1e-3) / 8e12, 3), l2_hit_rate=round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3) if "TCC_HIT_sum" in c else None, bit_exact=t["bit_exact"]))
json.dump(dict(what="battlefield-synth at four sizes, 1M incoherent rays (synth.random_rays seed 7), default kernel, one launch alone; traffic = 2 x FETCH_SIZE + WRITE_SIZE", rows=rows), open(out + "/summary.json", "w"), indent=1)
for r in rows: print(json.dumps(r))
PY
