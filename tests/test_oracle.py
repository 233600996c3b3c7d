"""CPU tests of the oracle itself (oracle/racc_oracle.c).

The reference has NO tests, golden vectors or fixtures for this path.  The oracle is pinned to
the reference itself on the GPU box (tests/test_gpu_reference_kernel.py runs the reference's own OpenCL kernel from
oracle/_ref); on CPU-only hosts these tests hold it to: (1) an independent double-precision brute-force arbiter over
all triangles, (2) hand-made known-answer cases for every quirk SURVEY.md §8(c) lists, (3) self-generated golden
vectors under tests/golden/ (regression protection; generator: tools/make_golden.py).
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from rayaccel_amd import synth
from helpers import MISS, assert_matches_arbiter, book_rays, book_scene, comb_scene, far_scene, leaf_rays, leaf_scene, make_rays, sliver_scene

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def small_blobs(small_scene):
    return orc.build_scene(small_scene["vertices"], small_scene["indices"])


def test_record_sizes():
    assert orc.RAY_DTYPE.itemsize == 32 and orc.RESULT_DTYPE.itemsize == 16          # RayAccelerator.h:59-76
    assert orc.GPU_NODE_DTYPE.itemsize == 64 and orc.PAIR_DTYPE.itemsize == 48       # Scene.cpp:73-87
    assert orc.BVH2_NODE_DTYPE.itemsize == 48                                        # Bvh2.h:15-22


@pytest.mark.parametrize("kind", ["primary", "diffuse", "random"])
def test_oracle_matches_bruteforce(small_scene, small_blobs, kind):
    prim, _ = synth.primary_rays(small_scene["camera"], 128, 128)
    if kind == "primary":
        rays = prim
    elif kind == "diffuse":
        rays = synth.diffuse_bounce_rays(small_scene, prim, orc.traverse(small_blobs, prim), 8192)
    else:
        rays = synth.random_rays(6000, seed=7, extent=100.0, ymax=30.0)
    res = orc.traverse(small_blobs, rays, env=small_scene["env"])
    assert (res["triangle"] != MISS).any() and (res["triangle"] == MISS).any()
    assert_matches_arbiter(res, small_scene, rays)


def test_counters_and_algorithmic_bytes(small_scene, small_blobs):
    rays, _ = synth.primary_rays(small_scene["camera"], 128, 128)
    res, nv, npairs, depth = orc.traverse(small_blobs, rays, counters=True)
    assert nv.min() >= 1 and depth.max() < 64                     # root is always visited; reference stack is 64
    hits = int((res["triangle"] != MISS).sum())
    assert orc.algorithmic_bytes(res, nv, npairs) == 48 * len(rays) + 64 * int(nv.sum()) + 48 * int(npairs.sum()) + 4 * hits
    res_mt = orc.traverse(small_blobs, rays, threads=4)
    assert np.array_equal(res_mt.view(np.uint8), orc.traverse(small_blobs, rays).view(np.uint8))


# ---------------------------------------------------------------- known-answer cases
def _quad_scene(extra=0):
    """Unit quad in z=0 made of triangles 0:(0,1,2) 1:(0,2,3) + a far dummy triangle so the root is inner."""
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [5, 5, 9], [6, 5, 9], [5, 6, 9]], np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6]], np.uint32)
    return dict(vertices=np.concatenate([v, np.ones((len(v), 1), np.float32)], 1), indices=idx)


def test_kat_barycentrics_follow_original_vertex_order():
    sc = _quad_scene()
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    assert blobs["pair_count"] == 2                                # the quad merged into one pair + the lone triangle
    rays = make_rays([[0.75, 0.25, -2], [0.25, 0.75, -2]], [[0, 0, 1], [0, 0, 1]])
    res = orc.traverse(blobs, rays)
    # tri 0 = (v0,v1,v2): P = v0 + u (v1-v0) + v (v2-v0) -> (0.75,0.25) => u=0.5, v=0.25
    assert res["triangle"][0] == 0 and res["t"][0] == 2.0 and abs(res["u"][0] - 0.5) < 1e-6 and abs(res["v"][0] - 0.25) < 1e-6
    # tri 1 = (v0,v2,v3): (0.25,0.75) = u (1,1) + v (0,1) => u=0.25, v=0.5
    assert res["triangle"][1] == 1 and abs(res["u"][1] - 0.25) < 1e-6 and abs(res["v"][1] - 0.5) < 1e-6


def test_kat_backface_and_interval():
    sc = _quad_scene()
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    back = orc.traverse(blobs, make_rays([[0.75, 0.25, 3]], [[0, 0, -1]]))
    assert back["triangle"][0] == 0 and back["t"][0] == 3.0          # sign-flip path (Kernels.h:60-66,85)
    at_min = orc.traverse(blobs, make_rays([[0.75, 0.25, -2]], [[0, 0, 1]], min_t=2.0))
    assert at_min["triangle"][0] == MISS                             # open at minT: T <= absDet*tNear rejects
    # The pair test is closed at maxT (Kernels.h:88) ...
    v = np.array([[0, 0, 0], [2, 0, 2], [0, 2, 2], [5, 5, 9], [6, 5, 9], [5, 6, 9], [7, 7, 9], [8, 7, 9], [7, 8, 9]], np.float32)
    slanted = dict(vertices=np.concatenate([v, np.ones((9, 1), np.float32)], 1), indices=np.arange(9, dtype=np.uint32).reshape(3, 3))
    sb = orc.build_scene(slanted["vertices"], slanted["indices"])
    at_max = orc.traverse(sb, make_rays([[0.5, 0.5, -2]], [[0, 0, 1]], max_t=3.0))
    assert at_max["triangle"][0] == 0 and at_max["t"][0] == 3.0
    short = orc.traverse(sb, make_rays([[0.5, 0.5, -2]], [[0, 0, 1]], max_t=2.999))
    assert short["triangle"][0] == MISS
    # ... but a child box whose ENTRY distance equals tFar exactly reads as "missed" (the slab test returns
    # tFar as its miss sentinel, Kernels.h:131-134,190-194), so a flat axis-aligned quad AT maxT is culled.
    flat = orc.traverse(blobs, make_rays([[0.75, 0.25, -2]], [[0, 0, 1]], max_t=2.0))
    assert flat["triangle"][0] == MISS


def test_kat_shared_edge_and_degenerate_second_triangle():
    sc = _quad_scene()
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    diag = orc.traverse(blobs, make_rays([[0.5, 0.5, -1]], [[0, 0, 1]]))       # exactly on the shared diagonal
    assert diag["triangle"][0] in (0, 1) and diag["t"][0] == 1.0
    lone = orc.traverse(blobs, make_rays([[5.25, 5.25, 0]], [[0, 0, 1]]))      # unpaired triangle: p3 = p1 => n2 = 0
    assert lone["triangle"][0] == 2 and lone["t"][0] == 9.0
    outside = orc.traverse(blobs, make_rays([[5.9, 5.9, 0]], [[0, 0, 1]]))     # would hit the phantom 2nd triangle if it existed
    assert outside["triangle"][0] == MISS


def test_kat_axis_parallel_direction_is_clamped():
    sc = _quad_scene()
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    res = orc.traverse(blobs, make_rays([[0.75, 0.25, -2], [0.75, 0.25, -2]], [[0, 0, 1], [-0.0, 0.0, 1]]))
    assert list(res["triangle"]) == [0, 0] and res["t"][0] == 2.0              # dir components 0 -> +-1e-10 (Kernels.h:149-157)


def test_kat_edge_codes_rotate_barycentrics():
    """A triangle reached through edge 1 / edge 2 of its pair must still report (u,v) in ORIGINAL vertex
    order (Kernels.h:223-239, Scene.cpp:132-136)."""
    v = np.array([[0, 0, 0], [2, 0, 0], [0, 2, 0], [2, 2, 0], [7, 7, 5], [8, 7, 5], [7, 8, 5]], np.float32)
    vv = np.concatenate([v, np.ones((len(v), 1), np.float32)], 1)
    for rot in range(3):
        a = np.roll(np.array([0, 1, 2]), rot)
        b = np.roll(np.array([1, 3, 2]), rot)
        idx = np.array([a, b, [4, 5, 6]], np.uint32)
        blobs = orc.build_scene(vv, idx)
        rays = make_rays([[0.5, 0.5, -1], [1.5, 1.5, -1]], [[0, 0, 1], [0, 0, 1]])
        res = orc.traverse(blobs, rays)
        tri, t, u, w, _ = orc.brute_closest(vv, idx, rays)
        assert np.array_equal(res["triangle"], tri)
        np.testing.assert_allclose(res["u"], u, atol=1e-6)
        np.testing.assert_allclose(res["v"], w, atol=1e-6)
    edges = set()
    for rot in range(3):
        idx = np.array([np.roll([0, 1, 2], rot), np.roll([1, 3, 2], (rot + 1) % 3), [4, 5, 6]], np.uint32)
        edges |= set((orc.build_scene(vv, idx)["remap"] >> 30).tolist())
    assert {1, 2, 3} <= edges                                                   # edge code 3 = (edge1+1) with edge1 = 2


def test_kat_leaf_with_126_triangles():
    """126 triangles that all have the same bounding box cannot be separated by the sweep -> ONE leaf of 126 triangles (the estimate says leaf,
    and only from 127 triangles on is a split forced: Bvh2.cpp:462-485) = 126 lone pairs; with 127 the forced median split gives 63 + 64.
    (Until round 5 this test built slivers with coincident centroids, which the sweep does separate: its largest leaf held 11 pairs.)"""
    sc = book_scene(126)
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    leaf_sizes = sorted((c >> 24) for c in np.concatenate([blobs["nodes"]["first"], blobs["nodes"]["last"]]) if not c & 0x80000000)
    assert leaf_sizes[-1] == 126
    rays = book_rays(sc)
    res, _, npairs, _ = orc.traverse(blobs, rays, counters=True)
    assert npairs.max() == 126 and (res["triangle"] != MISS).sum() >= len(rays) - 2
    assert_matches_arbiter(res, sc, rays)
    forced = book_scene(127)
    fb = orc.build_scene(forced["vertices"], forced["indices"])
    sizes = sorted((c >> 24) for c in np.concatenate([fb["nodes"]["first"], fb["nodes"]["last"]]) if not c & 0x80000000)
    assert sizes[-2:] == [63, 64]
    assert_matches_arbiter(orc.traverse(fb, book_rays(forced, 127)), forced, book_rays(forced, 127))


# ------------------------------------------------- the hand-made scenes of tests/test_gpu_edge_cases.py, oracle vs arbiter (CPU)
@pytest.mark.parametrize("arrangement", ["row", "near_first", "near_last", "coincident"])
@pytest.mark.parametrize("n_pairs", [1, 7, 63, 127])
def test_hand_made_leaves_match_the_arbiter(n_pairs, arrangement):
    """A leaf of up to 127 pairs (Scene.cpp:294-312) with truthful edge codes: the oracle's (u, v) after the remap rotation
    (Kernels.h:223-239) are the arbiter's, which only knows the triangles; of n coincident pairs the LAST one tested is kept (Kernels.h:88)."""
    blobs, geometry = leaf_scene(n_pairs, arrangement)
    rays = leaf_rays(n_pairs, arrangement)
    res, nv, npairs, _ = orc.traverse(blobs, rays, counters=True)
    assert_matches_arbiter(res, geometry, rays)
    assert npairs.max() == n_pairs and (nv == 1).all()
    if n_pairs >= 7:
        assert set((blobs["remap"] >> 30).tolist()) == {0, 1, 2, 3}
    if arrangement == "coincident":
        assert res["triangle"][0] // 2 == n_pairs - 1


def test_large_coordinates_slivers_and_the_other_scene_classes_match_the_arbiter():
    far = far_scene()
    prim, _ = synth.primary_rays(far["camera"], 128, 128)
    assert np.abs(far["vertices"][:, :3]).max() > 1e5 and np.median(np.abs(far["vertices"][:, :3])) > 1e4
    assert_matches_arbiter(orc.traverse(orc.build_scene(far["vertices"], far["indices"]), prim), far, prim)
    sl = sliver_scene()
    rays = synth.random_rays(20000, seed=3, extent=50.0, ymax=50.0)
    rays["origin"][:, 1] -= 25
    res = orc.traverse(orc.build_scene(sl["vertices"], sl["indices"]), rays)
    assert (res["triangle"] != MISS).sum() > 100
    assert_matches_arbiter(res, sl, rays, rel=5e-3, uv_atol=1e-2)      # needles of width 1e-6: what the binary32 pair test is good for
    for sc, min_leaf in ((synth.soup_synth(triangles=12000, clusters=24), 6), (synth.city_synth(blocks=8), 3)):
        blobs = orc.build_scene(sc["vertices"], sc["indices"])
        kids = np.concatenate([blobs["nodes"]["first"], blobs["nodes"]["last"]])
        assert (kids[kids & 0x80000000 == 0] >> 24).max() >= min_leaf
        prim, _ = synth.primary_rays(sc["camera"], 128, 128)
        hits = orc.traverse(blobs, prim)
        rays = np.concatenate([prim[::5], synth.diffuse_bounce_rays(sc, prim, hits, 3000)])
        assert_matches_arbiter(orc.traverse(blobs, rays), sc, rays, uv_atol=1e-4)


@pytest.mark.parametrize("width", [8, 16])
def test_simd_legs_are_the_scalar_port_bit_for_bit(small_scene, small_blobs, width):
    """oracle/racc_oracle_simd.c and racc_oracle_simd512.c (bench.py's cpu_baseline kind "simd-port": eight rays at a time in AVX2, as
    Scene.cpp:386-428 hands them to rtcIntersect8, or sixteen in AVX-512): every lane does what the scalar port does for its ray — byte-identical records on primaries, bounces, random rays,
    invalid and unbounded rays, with and without a probe image, on both trees, on leaves of 127 pairs (incl. 127 exact-distance ties) and
    on the 40-deep comb, for any thread count."""
    if orc.simd_width() < width:
        pytest.skip("host without " + ("AVX2 + FMA" if width == 8 else "AVX-512F/DQ/VL"))
    import rayaccel_amd as ra
    prim, _ = synth.primary_rays(small_scene["camera"], 256, 256)
    rays = np.concatenate([prim, synth.diffuse_bounce_rays(small_scene, prim, orc.traverse(small_blobs, prim), 30001), synth.random_rays(20003, seed=7, extent=100.0, ymax=30.0)])
    rays["dir"][5, 1] = np.nan; rays["origin"][77, 0] = np.inf; rays["maxT"][300:400] = np.inf; rays["dir"][401] = (0.0, -0.0, 1.0); rays["minT"][130] = -np.inf
    quality = ra.HostScene(small_scene["vertices"], small_scene["indices"], quality=1).blobs()
    for blobs in (small_blobs, quality):
        for env in (None, small_scene["env"]):
            want = orc.traverse(blobs, rays, env=env)
            for threads in (1, 3):
                assert orc.traverse_simd(blobs, rays, env=env, threads=threads, width=width).tobytes() == want.tobytes()
    for n in (1, 5, 17):       # fewer rays than lanes
        assert orc.traverse_simd(small_blobs, rays[:n], width=width).tobytes() == orc.traverse(small_blobs, rays[:n]).tobytes()
    assert len(orc.traverse_simd(small_blobs, rays[:0], width=width)) == 0
    for arrangement in ("row", "near_last", "coincident"):
        blobs, _ = leaf_scene(127, arrangement)
        r = leaf_rays(127, arrangement)
        assert orc.traverse_simd(blobs, r, width=width).tobytes() == orc.traverse(blobs, r).tobytes()
    comb = comb_scene(40)
    r = make_rays(np.stack([np.linspace(-20, 20, 300), np.linspace(-15, 15, 300), np.full(300, -10.0)], 1), [[0, 0, 1]] * 300)
    assert orc.traverse_simd(comb, r, width=width).tobytes() == orc.traverse(comb, r).tobytes()


def test_deep_stack_comb():
    blobs = comb_scene(40)
    res, nv, npairs, depth = orc.traverse(blobs, make_rays([[0, 0, -10]], [[0, 0, 1]]), counters=True)
    assert depth[0] == 40 and nv[0] == 40 and npairs[0] == 41
    assert res["triangle"][0] == 40 and res["t"][0] == 70.0


def test_invalid_rays_are_misses(small_blobs):
    rays = make_rays([[0, 50, 0]] * 4, [[0, -1, 0]] * 4)
    rays["dir"][1, 0] = np.nan
    rays["origin"][2, 2] = np.inf
    rays["maxT"][3] = np.inf          # +inf maxT is a VALID ray
    res = orc.traverse(small_blobs, rays)
    assert res["triangle"][1] == MISS and res["triangle"][2] == MISS and res["t"][1] == 0.0
    assert res["triangle"][0] == res["triangle"][3] != MISS and res["t"][0] == res["t"][3]


def test_env_sample_bilinear_clamp():
    env = np.zeros((4, 8, 4), np.float32)
    env[..., 0] = np.arange(8)[None, :]
    env[..., 1] = np.arange(4)[:, None]
    rgb = orc.env_sample(env, [[-1.0, 0.0, 0.0]])[0]          # acos(1)=0 -> r=0 -> centre of the image
    assert abs(rgb[0] - 3.5) < 1e-6 and abs(rgb[1] - 1.5) < 1e-6
    rgb = orc.env_sample(env, [[1.0, 1e-12, 0.0]])[0]          # rlen > 1e6 -> r = 0 (Kernels.h:217)
    assert abs(rgb[0] - 3.5) < 1e-6
    far = orc.env_sample(env, [[0.9999, 0.0, 0.0141]])[0]      # r ~ 0.5/|dz|... lands outside -> clamped to an edge texel
    assert 0.0 <= far[0] <= 7.0 and 0.0 <= far[1] <= 3.0


# ------------------------------------------------------------------------------ golden vectors
def test_golden_vectors():
    """Self-generated (tools/make_golden.py) — the reference holds no vectors for this path."""
    meta = json.load(open(os.path.join(GOLDEN, "golden_small.json")))
    data = np.load(os.path.join(GOLDEN, "golden_small.npz"))
    blobs = dict(nodes=data["nodes"].view(orc.GPU_NODE_DTYPE).reshape(-1), pairs=data["pairs"].view(orc.PAIR_DTYPE).reshape(-1),
                 remap=data["remap"], pair_count=int(meta["pair_count"]))
    rays = data["rays"].view(orc.RAY_DTYPE).reshape(-1)
    want = data["results"].view(orc.RESULT_DTYPE).reshape(-1)
    got, nv, npairs, _ = orc.traverse(blobs, rays, env=data["env"], counters=True)
    assert np.array_equal(got["triangle"], want["triangle"])
    hit = want["triangle"] != MISS
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit].view(np.uint32), want[f][hit].view(np.uint32))
        np.testing.assert_allclose(got[f][~hit], want[f][~hit], rtol=1e-5, atol=1e-5)
    assert orc.algorithmic_bytes(got, nv, npairs) == meta["algorithmic_bytes"]
    # the committed blobs are what today's builder makes from the committed mesh
    rebuilt = orc.build_scene(data["vertices"], data["indices"])
    assert np.array_equal(rebuilt["nodes"].view(np.uint8), blobs["nodes"].view(np.uint8))
    assert np.array_equal(rebuilt["pairs"].view(np.uint8), blobs["pairs"].view(np.uint8))
    assert np.array_equal(rebuilt["remap"], blobs["remap"])


def test_embree_adapter_plumbing_against_an_api_mock(tmp_path, small_scene, monkeypatch):
    """The optional system-Embree adapter (oracle/embree_adapter.py + embree_shim.c) had never EXECUTED in five rounds: no box has an
    Embree.  tests/cpp/embree_api_mock.c exports the twelve Embree 3/4 entry points the shim binds (behind them: a double-precision brute force —
    not Embree, pinning nothing about Embree's numerics).  With RACC_EMBREE_LIB pointing at it the adapter runs end to end: every dlsym, the version
    probe, the geometry buffers, the RTCRayHit fields, the threaded slice loop, the result conversion — and returns what the arbiter returns."""
    import subprocess
    from oracle import embree_adapter
    lib = os.path.join(str(tmp_path), "libembree_api_mock.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-shared", "-fPIC", "-Wall", os.path.join(os.path.dirname(__file__), "cpp", "embree_api_mock.c"), "-o", lib, "-lm"])
    monkeypatch.setenv("RACC_EMBREE_LIB", lib)
    assert embree_adapter.available() and embree_adapter.find_library() == lib
    sc = synth.battlefield_synth(grid=12, boxes=6, quads=20)      # ~400 triangles: the mock is a brute force
    prim, _ = synth.primary_rays(sc["camera"], 128, 128)
    rays = np.concatenate([prim[::7], synth.random_rays(1500, seed=3, extent=100.0, ymax=30.0)])
    got = embree_adapter.trace(sc, rays, threads=3)
    tri, t, u, v, _ = orc.brute_closest(sc["vertices"], sc["indices"], rays)
    assert np.array_equal(got["triangle"], tri) and (tri != MISS).sum() > 100 and (tri == MISS).sum() > 100
    hit = tri != MISS
    np.testing.assert_allclose(got["t"][hit], t[hit], rtol=1e-5)
    np.testing.assert_allclose(got["u"][hit], u[hit], atol=1e-5)
    np.testing.assert_allclose(got["v"][hit], v[hit], atol=1e-5)
    assert (got["t"][~hit] == 0).all()
    row = embree_adapter.time_batch(sc, rays[:500], 2)
    assert row["value"] > 0 and row["kind"] == "reference-dependency" and "Embree 4.x" in row["what"]


def test_embree_adapter_is_optional_and_its_shim_compiles():
    """oracle/embree_adapter.py: binds a system Embree when there is one — there is none on the boxes of this
    build, so all that can be checked is that the shim builds without any Embree header and that absence is reported."""
    from oracle import embree_adapter
    lib = embree_adapter.build_shim(force=True)
    assert hasattr(lib, "shim_open") and hasattr(lib, "shim_trace")
    if not embree_adapter.available():
        assert embree_adapter.find_library() is None
