"""CPU tests of the drop-in boundary: libracc_hip.so loads and exports exactly what include/racc_hip.h
declares; no compute entry point is callable without a GPU, and nothing falls back to the CPU."""
import ctypes as C
import os
import re

import pytest

import rayaccel_amd as ra
from rayaccel_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "racc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(racc_(?:hip|host)_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert len(names) >= 25
    lib = C.CDLL(engine.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "header declares %s but libracc_hip.so does not export it" % n
    assert sorted(engine.ABI) == names, "python binding and header disagree"
    assert ra.load_library().racc_hip_version().startswith(b"racc-hip")


def test_nothing_but_the_c_abi_is_exported():
    """Round-3 verdict: `rccl`, `failRccl`, `groupCollect` left libracc_hip.so unprefixed (helpers inside `extern "C"`).  The link
    step now carries an export list (rayaccel_amd/csrc/racc_hip.map): every dynamic symbol the library defines is a racc_hip_* /
    racc_host_* entry — no C++ template instantiations, no helper a host program's own `rccl` could collide with."""
    out = os.popen("nm -D --defined-only %s" % engine.LIB_PATH).read().split("\n")
    names = [l.split()[-1] for l in out if l.strip()]
    assert len(names) >= 50
    stray = [n for n in names if not re.match(r"racc_(hip|host)_[a-z0-9_]+$", n)]
    assert not stray, stray
    assert set(_declared()) <= set(names)


def test_struct_layouts_match_header():
    assert C.sizeof(engine.Options) == 18 * 4
    assert C.sizeof(engine.SceneInfo) == 32 and C.sizeof(engine.LaunchInfo) == 20
    assert ra.RAY_DTYPE.itemsize == 32 and ra.RESULT_DTYPE.itemsize == 16


def test_library_links_no_oracle_and_no_torch():
    """The product must not route through the oracle or any CPU fallback."""
    out = os.popen("ldd %s" % engine.LIB_PATH).read()
    assert "oracle" not in out and "torch" not in out
    syms = os.popen("nm -D %s" % engine.LIB_PATH).read()
    assert "orc_" not in syms
    for root, _, files in os.walk(os.path.join(ROOT, "rayaccel_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(import oracle|from oracle)", src, flags=re.M), f
                assert not re.search(r"#include\s*[<\"][^>\"]*oracle", src) and "libracc_oracle" not in src, f


@pytest.mark.skipif(ra.device_count() > 0, reason="only meaningful on a host without a GPU")
def test_fails_loudly_without_gpu():
    with pytest.raises(ra.RaccError) as e:
        ra.Context(device=0)
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)


def test_consumer_libraries_load_and_export_their_entry_points():
    """The two path-tracing consumers (host callbacks / device kernels) are separate libraries over the boundary."""
    for path, sym in ((engine.PT_LIB_PATH, "racc_pt_render_file"), (engine.PTDEV_LIB_PATH, "racc_ptdev_render_file"),
                      (engine.API_LIB_PATH, None)):
        assert os.path.exists(path), path
        lib = C.CDLL(path)
        if sym:
            assert hasattr(lib, sym)
        out = os.popen("ldd %s" % path).read()
        assert "libracc_hip.so" in out and "oracle" not in out and "torch" not in out


def test_build_has_the_current_kernels_only():
    """racc_hip_variant_available needs no GPU: 0 (default) and the V8 / V9 / V10 rows exist; the earlier generations (rows 1-40, retired
    from the tree in round 5) do not."""
    import rayaccel_amd as ra
    lib = ra.load_library()
    assert lib.racc_hip_variant_available(0) == 1 and lib.racc_hip_variant_available(41) == 1 and lib.racc_hip_variant_available(43) == 1
    assert lib.racc_hip_variant_available(1000) == 0 and lib.racc_hip_variant_available(22) == 0
    assert b"experimental" not in lib.racc_hip_version()
    assert ra.engine.Options.__dict__ is not None and __import__("ctypes").sizeof(ra.engine.Options) == 72      # the options block is ABI: 18 words (struct_size tells older callers apart)


def test_traversal_kernels_use_no_scratch():
    """Round 3's finding: with the ray constants as plain inputs of the assembly block the register allocator spilled
    a copy of them around it — scratch stores per lane per refill that made 7.7 x the compulsory write traffic.  No shipped traversal
    kernel may need scratch again, and the default ones must keep five waves per SIMD."""
    import re
    import subprocess
    from rayaccel_amd import engine
    out = subprocess.run(["make", "-s", "-C", engine.CSRC, "resources"], capture_output=True, text=True, timeout=600)
    text = out.stdout + out.stderr
    rows = re.findall(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+)", text, re.S)
    spills = {name: int(sp) for name, _, _, sp in rows if "traverseKernel" in name}
    kernels = {name: (int(scratch), int(occ)) for name, scratch, occ, _ in rows if "traverseKernel" in name}
    assert len(kernels) >= 15, text[-2000:]
    # (the one exception: kernel_variant 70, the six-waves-per-SIMD A/B of round 6 — 80 VGPRs leave its cold code 5-8 spilled registers; it is
    #  measured slower and never the default: DESIGN.md §3.  Its instantiations end in the template argument NARROW = true)
    narrow = lambda name: "traverseKernelV8ILi256ELi11" in name and name.endswith("Li0ELb1EEEvNS_12TraverseArgsE")
    assert sum(1 for k in kernels if narrow(k)) == 3 and all(occ == 6 for k, (_, occ) in kernels.items() if narrow(k))
    assert all(v == 0 for k, v in spills.items() if not narrow(k)), {k: v for k, v in spills.items() if v and not narrow(k)}      # no VGPR ever goes to scratch
    # ... and no private segment at all, except an untouched 36-byte frame the compiler leaves in the lazy-chain instantiation of the compressed
    # 4-wide kernel (SGPR spill slots that ended up in VGPR lanes: the kernel's code contains no scratch instruction)
    phantom = lambda name: "traverseKernelV10ILi256" in name and name.endswith("ELb1ELb1ELb1EEEvNS_12TraverseArgsE")
    assert all(s == 0 for k, (s, _) in kernels.items() if not narrow(k) and not phantom(k)), {k: v for k, v in kernels.items() if v[0] and not narrow(k)}
    assert all(s <= 64 for k, (s, _) in kernels.items() if phantom(k))
    for name, (_, occ) in kernels.items():
        if ("traverseKernelV8ILi256ELi13" in name and name.endswith("Li0ELb0EEEvNS_12TraverseArgsE")) or "traverseKernelV10ILi256ELi15ELb0ELb1" in name:      # (Li0: no LDS node cache — variants 60-63 run four waves per SIMD by design)
            assert occ >= 5, (name, occ)
