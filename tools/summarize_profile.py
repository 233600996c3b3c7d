"""Turns the rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into the committed summaries:
  profiles/<tag>/kernel_stats.csv            rocprofv3 --kernel-trace --stats of bench.py as the driver runs it (lanes overlap)
  profiles/<tag>/kernel_stats_one_lane.csv   the same command with one engine lane (every kernel alone on the GPU)
  profiles/<tag>/pmc_summary.json            mean of every counter over the 20 timed launches (PMC passes serialise kernels)
  profiles/<tag>/derived.json                what bench.py puts into its `roofline` object, each figure with its formula

Launch order of `bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3` among traverseKernel dispatches: [0] primary
batch through the host-buffer path, [1..3] warm-up, [4..23] the 20 timed diffuse launches."""
import collections, csv, glob, hashlib, json, os, shutil, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUS, SIMDS, XCDS = 256, 1024, 8
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    return f[-1] if f else None


def trace_durations(sub):
    t = newest(os.path.join(src, sub, "*", "*_kernel_trace.csv"))
    if not t:
        return None, None
    rows = [r for r in csv.DictReader(open(t)) if "traverseKernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    span = (max(int(r["End_Timestamp"]) for r in rows[4:24]) - min(int(r["Start_Timestamp"]) for r in rows[4:24])) / 1e6
    return rows, dict(kernel=rows[0]["Kernel_Name"], timed_mean_ms=float(np.mean(dur[4:24])), timed_min_ms=float(np.min(dur[4:24])),
                      timed_span_ms_per_launch=span / 20.0, launches=len(rows),
                      vgpr=rows[4].get("VGPR_Count"), sgpr=rows[4].get("SGPR_Count"), lds=rows[4].get("LDS_Block_Size"),
                      grid=rows[4].get("Grid_Size"), workgroup=rows[4].get("Workgroup_Size"))


out = {}
for sub, name in (("stats", "kernel_stats.csv"), ("stats_one_lane", "kernel_stats_one_lane.csv")):
    s = newest(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
    if s:
        shutil.copy(s, os.path.join(dst, name))
_, out["kernel_trace"] = trace_durations("stats")
_, out["kernel_trace_one_lane"] = trace_durations("stats_one_lane")
for d in sorted(filter(None, (newest(os.path.join(p, "*", "*_counter_collection.csv")) for p in glob.glob(os.path.join(src, "pmc_*"))))):   # newest run of every pass
    byc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "traverseKernel" in r["Kernel_Name"]:
            byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in byc.items():
        v.sort()
        vals = [x[1] for x in v]
        out[k] = dict(mean_timed_diffuse=float(np.mean(vals[4:24])), launches=len(vals))
h = hashlib.sha256()
for rel in ("rayaccel_amd/csrc/racc_kernel_v8.inc", "rayaccel_amd/csrc/racc_device.inc"):
    h.update(open(os.path.join(ROOT, rel), "rb").read())
out["kernel_source_sha256"] = h.hexdigest()
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)

g = lambda k: out.get(k, {}).get("mean_timed_diffuse")
cycles = g("GRBM_GUI_ACTIVE") / XCDS if g("GRBM_GUI_ACTIVE") else None          # the counter sums the 8 XCDs
one = out.get("kernel_trace_one_lane") or {}
der = dict(kernel_source_sha256=out["kernel_source_sha256"], source="%s/pmc_summary.json (rocprofv3 --pmc, one counter set per pass) + kernel_stats*.csv" % os.path.relpath(dst, ROOT),
           kernel=(out.get("kernel_trace") or {}).get("kernel"),
           kernel_ms_overlapped=(out.get("kernel_trace") or {}).get("timed_mean_ms"),
           ms_per_launch_overlapped=(out.get("kernel_trace") or {}).get("timed_span_ms_per_launch"),
           kernel_ms_isolated=one.get("timed_mean_ms"), kernel_cycles_isolated=cycles,
           formulas=dict(
               hbm_bytes_per_launch="2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024   (gfx950 tallies 128 B fetches at 64 B: MI355X_MICROARCH.md, HBM section)",
               hbm_physical_frac_isolated="hbm_bytes_per_launch / kernel_ms_isolated / 8 TB/s",
               td_busy_frac="TD_TD_BUSY_sum / (256 CUs * GRBM_GUI_ACTIVE / 8 XCDs)",
               ta_busy_frac="TA_TA_BUSY_sum / (256 CUs * cycles)",
               valu_busy_frac="4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles)   (the counter is in quad-cycles)",
               issue_slot_frac="4 * SQ_ACTIVE_INST_ANY / (1024 SIMDs * cycles)",
               valu_lane_util="SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)",
               salu_share="SQ_INSTS_SALU / (SQ_INSTS_VALU + SQ_INSTS_SALU + SQ_INSTS_VMEM_RD + SQ_INSTS_VMEM_WR + SQ_INSTS_LDS + SQ_INSTS_SMEM)",
               l2_hit_rate="TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)"))
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    der["hbm_bytes_per_launch"] = int(2 * g("FETCH_SIZE") * 1024.0 + g("WRITE_SIZE") * 1024.0)
    if one.get("timed_mean_ms"):
        der["hbm_physical_frac_isolated"] = round(der["hbm_bytes_per_launch"] / (one["timed_mean_ms"] * 1e-3) / 8e12, 4)
if cycles:
    if g("TD_TD_BUSY_sum"): der["td_busy_frac"] = round(g("TD_TD_BUSY_sum") / (CUS * cycles), 4)
    if g("TA_TA_BUSY_sum"): der["ta_busy_frac"] = round(g("TA_TA_BUSY_sum") / (CUS * cycles), 4)
    if g("SQ_ACTIVE_INST_VALU"): der["valu_busy_frac"] = round(4 * g("SQ_ACTIVE_INST_VALU") / (SIMDS * cycles), 4)
    if g("SQ_ACTIVE_INST_ANY"): der["issue_slot_frac"] = round(4 * g("SQ_ACTIVE_INST_ANY") / (SIMDS * cycles), 4)
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    der["valu_lane_util"] = round(g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")), 4)
insts = [g(k) or 0.0 for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM")]
if g("SQ_INSTS_SALU") and sum(insts):
    der["salu_share"] = round(g("SQ_INSTS_SALU") / sum(insts), 4)
    der["valu_insts_per_ray"] = round(g("SQ_INSTS_VALU") / (1 << 20), 1)
if g("TCC_HIT_sum") and g("TCC_MISS_sum"):
    der["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
busiest = max(((der.get(k) or 0.0, k) for k in ("td_busy_frac", "ta_busy_frac", "valu_busy_frac")), default=(0, None))
names = dict(td_busy_frac="TD (vector-memory data-return path of the CU)", ta_busy_frac="TA (vector-memory address path of the CU)", valu_busy_frac="VALU issue")
der["bound"] = "%s: busy %.0f %% of the launch, drain included" % (names.get(busiest[1], "?"), 100 * busiest[0]) if busiest[1] else None
json.dump(der, open(os.path.join(dst, "derived.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in der.items() if k != "formulas"}, indent=1))
