"""Turns the rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into the committed summaries:
  profiles/<tag>/kernel_stats.csv            rocprofv3 --kernel-trace --stats of bench.py as the driver runs it (lanes overlap)
  profiles/<tag>/kernel_stats_one_lane.csv   the same command with one engine lane (every kernel alone on the GPU)
  profiles/<tag>/pmc_summary.json            mean of every counter over the 20 timed launches (PMC passes serialise kernels)
  profiles/<tag>/derived.json                what bench.py puts into its `roofline` object, each figure with its formula

Launch order of `bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3` among traverseKernel dispatches: the primary batch
through the host-buffer path, (isolated launches unless RACC_BENCH_ISO_LAUNCHES=0 — tools/profile_bench.sh sets it), 3 warm-up steps,
the 20 timed launches: the LAST 20 traversal dispatches of a pass, which is how they are selected here (round 3 took rows 4..23 and
landed inside the isolated block: ADVICE r03)."""
import collections, csv, glob, hashlib, json, os, shutil, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUS, SIMDS, XCDS = 256, 1024, 8
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)


TIMED = {}      # per kernel-trace pass: the durations (ms) of the 20 timed traversal dispatches the means below are taken over


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
    if len(rows) < 20:
        if sub != "stats" or len(rows) < 3:
            return rows, None
        # the bench command as the driver runs it, lazily chained (round 5): the 20 timed steps are traced by the chain's kernels — one per
        # lane in rotation, the last three traversal dispatches of the pass — and nothing else; their common span is the timed region
        timed = rows[-3:]
        span = (max(int(r["End_Timestamp"]) for r in timed) - min(int(r["Start_Timestamp"]) for r in timed)) / 1e6
        TIMED[sub] = [round(d, 5) for d in dur[-3:]]
        return rows, dict(kernel=timed[0]["Kernel_Name"], timed_mean_ms=float(np.mean(dur[-3:])), timed_min_ms=float(np.min(dur[-3:])),
                          timed_span_ms_per_launch=span / 20.0, launches=len(rows), chain_kernels_in_timed_region=3,
                          vgpr=timed[0].get("VGPR_Count"), sgpr=timed[0].get("SGPR_Count"), lds=timed[0].get("LDS_Block_Size"),
                          grid=timed[0].get("Grid_Size"), workgroup=timed[0].get("Workgroup_Size"))
    timed = rows[-20:]
    TIMED[sub] = [round(d, 5) for d in dur[-20:]]
    span = (max(int(r["End_Timestamp"]) for r in timed) - min(int(r["Start_Timestamp"]) for r in timed)) / 1e6
    return rows, dict(kernel=timed[0]["Kernel_Name"], timed_mean_ms=float(np.mean(dur[-20:])), timed_min_ms=float(np.min(dur[-20:])),
                      timed_span_ms_per_launch=span / 20.0, launches=len(rows),
                      vgpr=timed[0].get("VGPR_Count"), sgpr=timed[0].get("SGPR_Count"), lds=timed[0].get("LDS_Block_Size"),
                      grid=timed[0].get("Grid_Size"), workgroup=timed[0].get("Workgroup_Size"))


def counters(prefix, sum_timed=False):
    """Mean over the 20 timed launches (the last 20 traversal dispatches of a pass) of every counter of the passes <prefix>_<i>;
    sum_timed: the SUM over those dispatches / 20 (chained launches: one kernel may do several batches' work, the others none)."""
    res = {}
    for d in sorted(filter(None, (newest(os.path.join(p, "*", "*_counter_collection.csv")) for p in glob.glob(os.path.join(src, prefix + "_[0-9]*"))))):
        byc = collections.defaultdict(list)
        for r in csv.DictReader(open(d)):
            if "traverseKernel" in r["Kernel_Name"]:
                byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, v in byc.items():
            v.sort()
            vals = [x[1] for x in v]
            if len(vals) < 20:
                if sum_timed and len(vals) >= 3:      # lazily chained: the timed steps are traced by the chain's three kernels, the last three dispatches
                    res[k] = dict(mean_timed=float(np.sum(vals[-3:])) / 20.0, launches=len(vals), chain_kernels=3)
                continue
            res[k] = dict(mean_timed=(float(np.sum(vals[-20:])) / 20.0 if sum_timed else float(np.mean(vals[-20:]))), launches=len(vals))
    return res


def derive(c, trace, rays=1 << 20):
    """The figures bench.py reports for one (kernel, batch) pair from its isolated PMC passes and its one-lane kernel trace."""
    g = lambda k: c.get(k, {}).get("mean_timed")
    cycles = g("GRBM_GUI_ACTIVE") / XCDS if g("GRBM_GUI_ACTIVE") else None          # the counter sums the 8 XCDs
    der = dict(kernel=(trace or {}).get("kernel"), kernel_ms_isolated=(trace or {}).get("timed_mean_ms"), kernel_cycles_isolated=cycles)
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        der["fabric_bytes_per_launch"] = int(2 * g("FETCH_SIZE") * 1024.0 + g("WRITE_SIZE") * 1024.0)
        der["fetch_bytes_per_launch"] = int(2 * g("FETCH_SIZE") * 1024.0)
        der["write_bytes_per_launch"] = int(g("WRITE_SIZE") * 1024.0)
        der["write_x_compulsory"] = round(g("WRITE_SIZE") * 1024.0 / (rays * 16.0), 2)
        if g("TCC_EA0_WRREQ_sum") is not None and g("TCC_EA0_WRREQ_64B_sum") is not None:
            # where the write traffic beyond the 16 B records comes from: the L2 writes back in 32 B and 64 B requests; a 16 B result that
            # leaves the L2 alone still costs a 32 B request, and a line whose records retire at different times is written back more than once
            w64, wall = g("TCC_EA0_WRREQ_64B_sum"), g("TCC_EA0_WRREQ_sum")
            der["write_requests"] = dict(total=wall, of_64B=w64, of_32B=wall - w64, bytes_by_request_size=int(64 * w64 + 32 * (wall - w64)),
                                         results_per_request=round(rays / wall, 3) if wall else None)
        if der["kernel_ms_isolated"]:
            der["fabric_frac_of_hbm_peak_isolated"] = round(der["fabric_bytes_per_launch"] / (der["kernel_ms_isolated"] * 1e-3) / 8e12, 4)
    if cycles:
        if g("TD_TD_BUSY_sum"): der["td_busy_frac"] = round(g("TD_TD_BUSY_sum") / (CUS * cycles), 4)
        if g("TA_TA_BUSY_sum"): der["ta_busy_frac"] = round(g("TA_TA_BUSY_sum") / (CUS * cycles), 4)
        if g("SQ_ACTIVE_INST_VALU"): der["valu_busy_frac"] = round(4 * g("SQ_ACTIVE_INST_VALU") / (SIMDS * cycles), 4)
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        der["valu_lane_util"] = round(g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")), 4)
    insts = [g(k) or 0.0 for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM")]
    if g("SQ_INSTS_SALU") and sum(insts):
        der["salu_share"] = round(g("SQ_INSTS_SALU") / sum(insts), 4)
        der["valu_insts_per_ray"] = round(g("SQ_INSTS_VALU") / rays, 2)
        der["vmem_rd_insts_per_ray"] = round(g("SQ_INSTS_VMEM_RD") / rays, 3)
    for k in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "TD_TD_BUSY_sum"):
        if g(k): der[k + "_per_launch"] = g(k)
    if g("TCC_HIT_sum") and g("TCC_MISS_sum"):
        der["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    # Where a wave's lifetime goes (the three states are disjoint and add up to SQ_WAVE_CYCLES: MI355X_MICROARCH.md, SQ counters): parked at an
    # s_waitcnt / barrier, stalled at issue, executing.  TD_TD_BUSY counts cycles in which the data-return path has a request OUTSTANDING — a
    # latency measure, not a volume one: taking a third of the L1 accesses away (the LDS node cache, profiles/r05/cache_ab_pmc.json) moves
    # neither it nor the launch time.
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") and g("SQ_ACTIVE_INST_ANY") and g("SQ_WAIT_INST_ANY") is not None:
        wc = g("SQ_WAVE_CYCLES")
        der["wave_time_split"] = dict(waiting=round(g("SQ_WAIT_ANY") / wc, 4), issue_stalled=round(g("SQ_WAIT_INST_ANY") / wc, 4), executing=round(g("SQ_ACTIVE_INST_ANY") / wc, 4))
    busiest = max(((der.get(k) or 0.0, k) for k in ("td_busy_frac", "ta_busy_frac", "valu_busy_frac")), default=(0, None))
    names = dict(td_busy_frac="TD (the CU's vector-memory data-return path has a request outstanding)", ta_busy_frac="TA (vector-memory address path of the CU)", valu_busy_frac="VALU issue")
    if busiest[0]:
        wts = der.get("wave_time_split")
        der["bound"] = ("latency of a step's dependent chain at this occupancy: a wave spends %.0f %% of its lifetime parked at a wait, %.0f %% stalled at issue, %.0f %% executing; busiest unit %s: %.0f %% of the launch, drain included"
                        % (100 * wts["waiting"], 100 * wts["issue_stalled"], 100 * wts["executing"], names.get(busiest[1], "?"), 100 * busiest[0])) if wts else \
                       "%s: busy %.0f %% of the launch, drain included" % (names.get(busiest[1], "?"), 100 * busiest[0])
    else:
        der["bound"] = None
    return der


out = {}
for sub, name in (("stats", "kernel_stats.csv"), ("stats_one_lane", "kernel_stats_one_lane.csv"), ("stats_one_lane_coherent", "kernel_stats_one_lane_coherent.csv"),
                  ("stats_one_lane_v10", "kernel_stats_one_lane_v10.csv"), ("stats_one_lane_xl", "kernel_stats_one_lane_xl.csv"),
                  ("stats_one_lane_xl_diffuse", "kernel_stats_one_lane_xl_diffuse.csv"), ("stats_one_lane_q0", "kernel_stats_one_lane_q0.csv"),
                  ("stats_one_lane_xl_q0", "kernel_stats_one_lane_xl_q0.csv")):
    s_ = newest(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
    if s_:
        shutil.copy(s_, os.path.join(dst, name))
_, out["kernel_trace"] = trace_durations("stats")
traces = {k: trace_durations(sub)[1] for k, sub in (("diffuse", "stats_one_lane"), ("coherent", "stats_one_lane_coherent"), ("v10_diffuse", "stats_one_lane_v10"),
                                                    ("xl", "stats_one_lane_xl"), ("xl_diffuse", "stats_one_lane_xl_diffuse"),
                                                    ("q0_diffuse", "stats_one_lane_q0"), ("q0_xl", "stats_one_lane_xl_q0"))}
out["kernel_trace_one_lane"] = traces
pm = {"diffuse": counters("pmc"), "coherent": counters("pmcc"), "v10_diffuse": counters("pmcv"), "diffuse_chained": counters("pmcx", sum_timed=True),
      "xl": counters("pmcxl"), "xl_diffuse": counters("pmcxd"), "q0_diffuse": counters("pmcq"), "q0_xl": counters("pmcxq")}
out["counters"] = pm
sys.path.insert(0, ROOT)
import bench      # the list of files the counters depend on lives there (KERNEL_SOURCES)
out["kernel_source_sha256"] = bench.kernel_source_sha256()
out["kernel_v10_source_sha256"] = hashlib.sha256(open(os.path.join(ROOT, "rayaccel_amd/csrc/racc_kernel_v10.inc"), "rb").read()).hexdigest()
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)

der = dict(kernel_source_sha256=out["kernel_source_sha256"], kernel_v10_source_sha256=out["kernel_v10_source_sha256"],
           source="%s/pmc_summary.json (rocprofv3 --pmc, one counter set per pass, the kernel alone on the GPU: one lane, no chaining) + kernel_stats*.csv" % os.path.relpath(dst, ROOT),
           kernel=(out.get("kernel_trace") or {}).get("kernel"),
           kernel_ms_overlapped=(out.get("kernel_trace") or {}).get("timed_mean_ms"),
           ms_per_launch_overlapped=(out.get("kernel_trace") or {}).get("timed_span_ms_per_launch"),
           formulas=dict(
               fabric_bytes_per_launch="2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024   (gfx950 tallies 128 B fetches at 64 B: MI355X_MICROARCH.md, HBM section); L2-miss traffic, Infinity-Cache hits included",
               fabric_frac_of_hbm_peak_isolated="fabric_bytes_per_launch / kernel_ms_isolated / 8 TB/s",
               write_x_compulsory="WRITE_SIZE * 1024 / (16 B x rays)",
               td_busy_frac="TD_TD_BUSY_sum / (256 CUs * GRBM_GUI_ACTIVE / 8 XCDs)",
               ta_busy_frac="TA_TA_BUSY_sum / (256 CUs * cycles)",
               valu_busy_frac="4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles)   (the counter is in quad-cycles)",
               valu_lane_util="SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)",
               salu_share="SQ_INSTS_SALU / (SQ_INSTS_VALU + SQ_INSTS_SALU + SQ_INSTS_VMEM_RD + SQ_INSTS_VMEM_WR + SQ_INSTS_LDS + SQ_INSTS_SMEM)",
               l2_hit_rate="TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)",
               vmem_rd_insts_per_ray="SQ_INSTS_VMEM_RD / 2^20 rays (wave-level instructions)",
               wave_time_split="SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, each / SQ_WAVE_CYCLES (disjoint states of a resident wave)",
               fabric_bytes_per_step_chained="(2 * sum FETCH_SIZE + sum WRITE_SIZE) * 1024 over the traversal dispatches of the 20 timed steps / 20, three lanes, chained (rocprofv3 serialises dispatches under --pmc)"))
for key in ("diffuse", "coherent", "v10_diffuse", "xl", "xl_diffuse", "q0_diffuse", "q0_xl"):
    der[key] = derive(pm[key], traces.get(key))
cx = pm["diffuse_chained"]
if cx.get("FETCH_SIZE") and cx.get("WRITE_SIZE"):
    der["diffuse"]["fabric_bytes_per_step_chained"] = int((2 * cx["FETCH_SIZE"]["mean_timed"] + cx["WRITE_SIZE"]["mean_timed"]) * 1024.0)
# round 6: the instantiation that is actually TIMED — lazily chained, misses shaded in the kernel — from the pmcx passes (default options).  Counters
# are per step (summed over the chain's kernels / 20: under --pmc rocprofv3 serialises dispatches, the chain's first kernel works through all 20
# batches); the kernel name and the per-step span come from the chained kernel trace (`stats`).
ct = out.get("kernel_trace") or {}
der["diffuse_chained"] = derive(cx, dict(kernel=ct.get("kernel"), timed_mean_ms=ct.get("timed_span_ms_per_launch")))
der["diffuse_chained"]["what"] = "per step of the lazily chained timed region (sum over the chain's kernels / 20); kernel_ms_isolated here = the span of the chain's kernels / 20 in the kernel-trace pass"
d0, d1 = der["diffuse"], der["v10_diffuse"]
if d0.get("SQ_INSTS_VMEM_RD_per_launch") and d1.get("SQ_INSTS_VMEM_RD_per_launch"):
    der["v10_vs_default"] = dict(vmem_rd_insts=round(d1["SQ_INSTS_VMEM_RD_per_launch"] / d0["SQ_INSTS_VMEM_RD_per_launch"], 3),
                                 td_busy_cycles=round(d1["TD_TD_BUSY_sum_per_launch"] / d0["TD_TD_BUSY_sum_per_launch"], 3) if d0.get("TD_TD_BUSY_sum_per_launch") and d1.get("TD_TD_BUSY_sum_per_launch") else None,
                                 valu_insts=round(d1["SQ_INSTS_VALU_per_launch"] / d0["SQ_INSTS_VALU_per_launch"], 3),
                                 kernel_ms=round(d1["kernel_ms_isolated"] / d0["kernel_ms_isolated"], 3) if d0.get("kernel_ms_isolated") and d1.get("kernel_ms_isolated") else None)
# copies of the microbenchmark outputs taken on the same box (tools/microbench/run_microbench.sh <tag>)
mb = os.path.join(ROOT, "gpurun_out", "microbench_" + tag)
for f in ("gather64.txt", "gather128.txt", "scatter16.txt", "fetchcal.txt", "fetchcal.json", "microbench.json"):
    if os.path.exists(os.path.join(mb, f)):
        shutil.copy(os.path.join(mb, f), os.path.join(dst, f))
der["tree"] = "diffuse / coherent / v10_diffuse / xl / xl_diffuse: racc_host_build_options.quality = 1 (bench.py's default); q0_diffuse / q0_xl: the reference builder's tree (--quality 0), same kernel, same rays"
# kernel_stats*.csv are rocprofv3's own aggregates over EVERY dispatch of a pass (the primary batch's slices and the warm-up steps included);
# the means in derived.json are over the 20 timed dispatches only — listed here so that they can be recomputed from a committed file
json.dump(dict(what="durations in ms of the last 20 traversal dispatches of each kernel-trace pass (= the 20 timed steps)", passes=TIMED),
          open(os.path.join(dst, "timed_dispatches.json"), "w"), indent=1)
for extra in ("step_stats.json", "steady_state_pmc.json"):
    if os.path.exists(os.path.join(ROOT, "gpurun_out", extra)):
        shutil.copy(os.path.join(ROOT, "gpurun_out", extra), os.path.join(dst, extra))
json.dump(der, open(os.path.join(dst, "derived.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in der.items() if k != "formulas"}, indent=1))
