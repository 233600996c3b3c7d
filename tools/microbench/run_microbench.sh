#!/bin/bash
# Runs the gather / scatter microbenchmarks on the GPU box and writes their raw output plus a JSON digest to gpurun_out/microbench_<tag>/
# (copied into profiles/<tag>/ afterwards: bench.py reads the gather ceiling from there).  Usage: tools/microbench/run_microbench.sh <tag>
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/microbench_$TAG
mkdir -p "$OUT"
cd "$REPO/tools/microbench"
for b in gather64 gather128 scatter16 fetchcal; do
  [ -x $b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value -o $b $b.hip || exit 1
done
./gather64 > "$OUT/gather64.txt" 2>&1
./gather128 > "$OUT/gather128.txt" 2>&1
./scatter16 > "$OUT/scatter16.txt" 2>&1
./fetchcal > "$OUT/fetchcal.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -k 5 120 rocprofv3 --pmc $c --output-format csv -d "$OUT/scatter_$c" -- "$REPO/tools/microbench/scatter16" > "$OUT/scatter_$c.log" 2>&1
done
# FETCH_SIZE calibration for the engine's own gather (fetchcal.hip): the counters per mode, one counter set per pass
for c in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_HIT_sum"; do
  d=$(echo $c | cut -d' ' -f1)
  timeout -k 5 180 rocprofv3 --pmc $c --output-format csv -d "$OUT/fetchcal_$d" -- "$REPO/tools/microbench/fetchcal" > "$OUT/fetchcal_$d.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, re, sys
out = sys.argv[1]
d = {"gather64": [], "gather128": [], "scatter16": {}}
for line in open(out + "/gather64.txt"):
    m = re.search(r"mode (\d+): ([\d.]+) ms, ([\d.]+) G record-gathers/s, ([\d.]+) cycles per wave-gather per CU", line)
    if m:
        d["gather64"].append({"line": line.strip(), "mode": int(m.group(1)), "ms": float(m.group(2)), "cycles_per_wave_gather_per_CU": float(m.group(4)),
                              "B_per_clk_per_CU": round(4096.0 / float(m.group(4)), 2)})
for line in open(out + "/gather128.txt"):
    m = re.search(r"table\s+([\d.]+) MB mode (\d+) \(\s*(\d+) B records\): ([\d.]+) ms, ([\d.]+) cycles per wave-gather per CU at 2.4 GHz = ([\d.]+) B/clk/CU", line)
    if m:
        d["gather128"].append({"table_MB": float(m.group(1)), "mode": int(m.group(2)), "record_B": int(m.group(3)), "ms": float(m.group(4)),
                               "cycles_per_wave_gather_per_CU": float(m.group(5)), "B_per_clk_per_CU": float(m.group(6))})
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    for f in glob.glob(out + "/scatter_%s/*/*_counter_collection.csv" % c):
        rows = sorted((int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f)) if "scatter" in r["Kernel_Name"] and r["Counter_Name"] == c)
        d["scatter16"][c + "_KB_by_mode"] = [v for _, v in rows]
d["scatter16"]["compulsory_KB"] = (1 << 20) * 16 / 1024.0
# fetchcal: per mode (second repetition of each), counters against the known line count
fc = {"lines_touched": 1 << 22, "line_bytes": 128, "modes": {"0": "streaming 4 KiB per wave", "1": "gather, one 64 B record per line", "2": "gather, both records of a line together",
                                                               "3": "gather, both records of a line a pass apart (line fetched twice)"}, "ms": {}, "counters": {}}
for line in open(out + "/fetchcal.txt"):
    m = re.search(r"mode (\d+): ([\d.]+) ms", line)
    if m:
        fc["ms"][m.group(1)] = float(m.group(2))
for f in glob.glob(out + "/fetchcal_*/*/*_counter_collection.csv"):
    by = {}
    for r in csv.DictReader(open(f)):
        if "fetchcal" in r["Kernel_Name"]:
            by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for name, rows in by.items():
        rows.sort()
        fc["counters"][name] = {str(mode): rows[2 * mode + 1][1] for mode in range(4) if len(rows) >= 2 * mode + 2}
if "FETCH_SIZE" in fc["counters"]:
    fc["FETCH_SIZE_bytes_tallied_per_line_touched"] = {m: round(v * 1024.0 / (1 << 22), 2) for m, v in fc["counters"]["FETCH_SIZE"].items()}
d["fetchcal"] = fc
best = [g for g in d["gather64"] if g["mode"] == 2]
if best:
    d["gather64_ceiling_B_per_clk_per_CU"] = max(g["B_per_clk_per_CU"] for g in best)
json.dump(d, open(out + "/microbench.json", "w"), indent=1)
json.dump(fc, open(out + "/fetchcal.json", "w"), indent=1)
print(json.dumps({k: v for k, v in d.items() if k not in ("gather64", "gather128")}, indent=1))
PY
