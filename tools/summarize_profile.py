"""Turns the rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into the committed summaries:
profiles/<tag>/kernel_stats.csv, profiles/<tag>/pmc_summary.json and profiles/traffic.json.

Launch order of `bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3` among traverseKernel dispatches: [0] primary
batch through the host-buffer path, [1..3] warm-up, [4..23] the 20 timed diffuse launches, then the extras."""
import collections, csv, glob, json, os, shutil, sys
import numpy as np

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", "prof_" + tag)
dst = os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)   # newest run if several were merged
if stats:
    shutil.copy(stats[-1], os.path.join(dst, "kernel_stats.csv"))
trace = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1:]
out = {}
if trace:
    rows = [r for r in csv.DictReader(open(trace[0])) if "traverseKernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    out["kernel_trace"] = dict(kernel=rows[0]["Kernel_Name"], timed_diffuse_mean_ms=float(np.mean(dur[4:24])), timed_diffuse_min_ms=float(np.min(dur[4:24])),
                               vgpr=rows[0].get("VGPR_Count"), sgpr=rows[0].get("SGPR_Count"), lds=rows[0].get("LDS_Block_Size"),
                               grid=rows[4].get("Grid_Size"), workgroup=rows[4].get("Workgroup_Size"))
_newest = {}
for d in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):   # newest file of every counter set
    k = d.split(os.sep)[-3]
    if k not in _newest or os.path.getmtime(d) > os.path.getmtime(_newest[k]):
        _newest[k] = d
for d in sorted(_newest.values()):
    byc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "traverseKernel" in r["Kernel_Name"]:
            byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in byc.items():
        v.sort()
        vals = [x[1] for x in v]
        out[k] = dict(mean_timed_diffuse=float(np.mean(vals[4:24])), launches=len(vals))
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
g = lambda k: out.get(k, {}).get("mean_timed_diffuse")
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    fetch_b, write_b = g("FETCH_SIZE") * 1024.0, g("WRITE_SIZE") * 1024.0
    traffic = dict(hbm_bytes_per_launch=int(2 * fetch_b + write_b), fetch_size_bytes_raw=int(fetch_b), write_size_bytes_raw=int(write_b),
                   correction="FETCH_SIZE doubled (gfx950 tallies 128 B requests at 64 B, MI355X_MICROARCH.md §HBM); WRITE_SIZE uncalibrated, taken as is",
                   source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3; mean of the 20 timed launches",
                   profile=dst, kernel=out.get("kernel_trace", {}).get("kernel"))
    if g("TCC_HIT_sum") and g("TCC_MISS_sum"):
        traffic["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    json.dump(traffic, open(os.path.join("profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
print(json.dumps(out.get("kernel_trace"), indent=1))
