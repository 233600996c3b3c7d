"""The committed rocprofv3 summaries must describe the kernel that is in the tree: profiles/r02/derived.json records the
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
    assert not prof["stale"], "kernel sources changed after %s was taken: re-profile (tools/profile_bench.sh r02)" % bench.PROFILE_DIR
    for key in ("hbm_bytes_per_launch", "td_busy_frac", "valu_busy_frac", "issue_slot_frac", "valu_lane_util", "salu_share", "kernel_ms_isolated"):
        assert prof.get(key) is not None and prof[key] > 0, key
    assert prof["hbm_physical_frac_isolated"] < 1 and prof["td_busy_frac"] <= 1 and prof["issue_slot_frac"] <= 1.01
    stats = open(os.path.join(ROOT, bench.PROFILE_DIR, "kernel_stats.csv")).read()
    assert bench.KERNEL_NAME in stats
