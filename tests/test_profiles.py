"""The committed rocprofv3 summaries must describe the kernel that is in the tree: profiles/<round>/derived.json records the
hash of the kernel sources it was taken with (tools/summarize_profile.py); bench.py reports numbers from it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_committed_profile_matches_the_kernel_sources():
    import bench
    prof = bench.committed_profile()
    assert prof is not None, "no %s/derived.json: run tools/profile_bench.sh + tools/summarize_profile.py" % bench.PROFILE_DIR
    assert not prof["stale"], "kernel sources changed after %s was taken: re-profile (tools/profile_bench.sh <round>)" % bench.PROFILE_DIR
    for wl in ("diffuse", "coherent"):        # configs[2] (the headline) and configs[1], each from its own isolated passes
        d = prof[wl]
        for key in ("fabric_bytes_per_launch", "td_busy_frac", "valu_busy_frac", "valu_lane_util", "salu_share", "kernel_ms_isolated"):
            assert d.get(key) is not None and d[key] > 0, (wl, key)
        assert d["fabric_frac_of_hbm_peak_isolated"] < 1 and d["td_busy_frac"] <= 1 and d["valu_busy_frac"] <= 1
    # round 6: the instantiation that is actually timed (lazily chained, in-kernel miss shading) and the fast-mode kernel have their own counters
    for section, kernel in (("diffuse_chained", "traverseKernelV8<256, 13, false, false, true, true"), ("v10_diffuse", "traverseKernelV10")):
        d = prof[section]
        assert kernel in d["kernel"], (section, d["kernel"])
        for key in ("fabric_bytes_per_launch", "td_busy_frac", "valu_busy_frac", "valu_lane_util", "vmem_rd_insts_per_ray", "valu_insts_per_ray", "l2_hit_rate", "kernel_ms_isolated"):
            assert d.get(key) is not None and d[key] > 0, (section, key)
        w = d["wave_time_split"]
        assert 0.95 < w["waiting"] + w["issue_stalled"] + w["executing"] < 1.02
    assert prof["v10_diffuse"]["vmem_rd_insts_per_ray"] < 0.75 * prof["diffuse"]["vmem_rd_insts_per_ray"]      # what the compressed 4-wide node is for
    assert bench.gather_ceiling() and 20 < bench.gather_ceiling() < 64          # measured, committed: profiles/<round>/microbench.json + gather64.txt
    assert os.path.exists(os.path.join(ROOT, bench.PROFILE_DIR, "gather64.txt"))
    stats = open(os.path.join(ROOT, bench.PROFILE_DIR, "kernel_stats.csv")).read()
    assert bench.KERNEL_NAME in stats


def test_steady_state_counters_match_the_kernel_sources():
    """profiles/<round>/steady_state_pmc.json (tools/steady_pmc.sh: the limiter counters of 8M-ray launches) is reported by bench.py as
    `roofline.limiter_steady_state`: same staleness rule, and the fractions are fractions."""
    import bench
    st = bench.steady_state_profile()
    assert st is not None and not st["stale"], "kernel sources changed after steady_state_pmc.json was taken: re-run tools/steady_pmc.sh"
    for key in ("valu_busy_frac", "valu_lane_util", "td_busy_frac", "ta_busy_frac", "l2_hit_rate"):
        assert 0 < st[key] <= 1, key
    w = st["wave_time_split"]
    assert 0.95 < w["waiting"] + w["issue_stalled"] + w["executing"] < 1.02      # disjoint states of a resident wave
    assert st["vmem_rd_insts_per_ray"] > 2 and st["valu_insts_per_ray"] > 40
