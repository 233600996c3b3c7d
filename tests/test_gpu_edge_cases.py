"""The reference's edge cases THROUGH THE HIP ENGINE, for every kernel instantiation that ships (round-5 verdict, item 1).

Until round 5 the hand-made known-answer cases of SURVEY.md §8(c) — open/closed ray interval (Kernels.h:88-89), the cross-multiplied
pick of the nearer triangle of a pair (Kernels.h:97), the degenerate second triangle (Scene.cpp:174-178), the edge codes incl. 3
(Scene.cpp:132-133), the 126-triangle leaf (Bvh2.cpp:467-485) — ran on the CPU oracle (tests/test_oracle.py) and on the reference's own
kernel (tests/test_gpu_reference_kernel.py) only, and no scene of the -m gpu suite held a leaf of more than six pairs (the format
allows 127, Scene.cpp:294-312).  Here every case is uploaded through racc_hip_scene_upload and traced by

  * the default kernel (V8) stand-alone (host path and chain_launches = 2), lazily chained (chain_min_rays = 1: batch after batch
    published to the running kernels, misses shaded by the traversal kernel) and chained with per-launch kernels (chain_launches = 3);
  * its 8-entry-stack, statistics, inline-probe and LDS-node-cache instantiations (kernel_variant 41, 42, 44, 60-63) and the narrow six-waves-per-SIMD one (70);
  * the 4-wide kernels (45, 46, 48, 49) and the compressed 4-wide ones (50, 51, 53), chained where they have such an instantiation;
  * the reference's own OpenCL kernel on the same blobs (oracle/_ref, where built).

Bar: the V8 family bit-exact against the oracle (primId, t, u, v; miss colours 1e-5); the wide kernels the oracle's closest hit
except exact-distance ties (tests/helpers.py::assert_same_closest_hit); the reference kernel north_star's 1e-4.  The scene info's
max_leaf_pairs is asserted, so every test says what it exercised."""
import numpy as np
import pytest

import rayaccel_amd as ra
from oracle import oracle as orc, ref_kernel
from rayaccel_amd import synth
from helpers import (MISS, QUANT_VARIANTS, WIDE_VARIANTS, assert_bit_exact, assert_matches_arbiter, assert_same_closest_hit,
                     book_rays, book_scene, compare_with_reference_kernel, far_scene, leaf_rays, leaf_scene, make_rays, sliver_scene)

pytestmark = pytest.mark.gpu

# name -> Context options.  chain_min_rays = 1 makes the device path chain batches of any size (the default threshold is 786,432 rays).
CONFIGS = {
    "v8, default options (small batches stand alone)": dict(),
    "v8 (default), lazily chained": dict(chain_min_rays=1),
    "v8 stand-alone": dict(chain_launches=2),
    "v8 chained, a kernel per launch": dict(chain_launches=3, chain_min_rays=1),
    "v8 8-entry stack (41)": dict(kernel_variant=41, chain_min_rays=1),
    "v8 statistics (42)": dict(kernel_variant=42),
    "v8 inline probe lookup (44)": dict(kernel_variant=44, chain_min_rays=1),
    "v8 LDS node cache 524 (60)": dict(kernel_variant=60, chain_min_rays=1),
    "v8 LDS node cache 768 (61)": dict(kernel_variant=61, chain_min_rays=1),
    "v8 LDS node cache 260 (62)": dict(kernel_variant=62, chain_min_rays=1),
    "v8 LDS node cache 128 (63)": dict(kernel_variant=63, chain_min_rays=1),
    "v8 narrow, six waves per SIMD (70)": dict(kernel_variant=70, chain_min_rays=1),
    "v8 narrow, stand-alone (70)": dict(kernel_variant=70, chain_launches=2),
    "v9 4-wide (45)": dict(kernel_variant=45),
    "v9 4-wide C++ 6-entry (46)": dict(kernel_variant=46),
    "v9 4-wide C++ (48)": dict(kernel_variant=48),
    "v9 4-wide 7-entry (49)": dict(kernel_variant=49),
    "v10 compressed 4-wide (50)": dict(kernel_variant=50, chain_min_rays=1),
    "v10 compressed 4-wide C++ (51)": dict(kernel_variant=51, chain_min_rays=1),
    "v10 compressed 4-wide 7-entry (53)": dict(kernel_variant=53, chain_min_rays=1),
}
ENV = synth.environment_synth(64, 32)


@pytest.fixture(scope="module")
def contexts():
    ctxs = {}
    for name, opt in CONFIGS.items():
        ctxs[name] = ra.Context(device=0, **opt)          # raises (never falls back) when the extension or the GPU is missing
    yield ctxs
    for c in ctxs.values():
        c.destroy()


def _device_path(ctx, scene, env, rays, batches=3):
    """racc_hip_intersect_device on the engine's own streams, `batches` times back to back into separate result arrays, then one wait:
    with chaining on, batch 2 and 3 are published to the kernels batch 1 started."""
    n = len(rays)
    d_r = ctx.alloc(n * 32)
    d_r.upload(rays)
    outs = [ctx.alloc(n * 16) for _ in range(batches)]
    for o in outs:
        ctx.intersect_device(scene, env, d_r.ptr, o.ptr, n, lane=ra.LANE_AUTO)
    ctx.wait(ra.LANE_AUTO)
    res = [o.download(orc.RESULT_DTYPE, n) for o in outs]
    for o in outs:
        o.free()
    d_r.free()
    return res


def run_everywhere(contexts, blobs, rays, what, env=ENV, min_leaf=None, arbiter=None, max_ties=None, reference_ties=None, skip_reference=False):
    """Uploads `blobs` into every context, traces `rays` through the host path and the device path, holds every result to the oracle —
    and the reference's own kernel on the same blobs to it as well.  Returns the oracle's records."""
    ref = orc.traverse(blobs, rays, env=env)
    for name, ctx in contexts.items():
        variant = CONFIGS[name].get("kernel_variant", 0)
        scene = ctx.upload_scene(blobs["nodes"], blobs["pairs"], blobs["remap"])
        if min_leaf is not None:
            assert scene.info["max_leaf_pairs"] >= min_leaf, "%s: the scene holds leaves of at most %d pairs, %d wanted" % (what, scene.info["max_leaf_pairs"], min_leaf)
        e = ctx.create_environment(env) if env is not None else None
        results = [("host path", ctx.intersect(scene, e, rays))] + [("device path, batch %d" % k, r) for k, r in enumerate(_device_path(ctx, scene, e, rays))]
        for path, got in results:
            label = "%s | %s | %s" % (what, name, path)
            if variant in WIDE_VARIANTS:
                arb = dict(arbiter, rays=rays) if (arbiter is not None and variant in QUANT_VARIANTS) else None
                assert_same_closest_hit(got, ref, label, max_ties=len(rays) if max_ties is None else max_ties, arbiter=arb)
            else:
                assert_bit_exact(got, ref, label)
        scene.destroy()
        if e is not None:
            e.destroy()
    if ref_kernel.built() and not skip_reference:
        compare_with_reference_kernel(ref_kernel.run(blobs, rays, env if env is not None else np.zeros((2, 2, 4), np.float32)), ref,
                                      "%s | reference kernel vs oracle" % what, max_ties=len(rays) if reference_ties is None else reference_ties)
    return ref


# --------------------------------------------------------------------------------------------------- the known-answer cases of tests/test_oracle.py
def _quad_scene():
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [5, 5, 9], [6, 5, 9], [5, 6, 9]], np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6]], np.uint32)
    return dict(vertices=np.concatenate([v, np.ones((len(v), 1), np.float32)], 1), indices=idx)


def test_kat_quad_barycentrics_backface_interval_shared_edge_degenerate_clamp(contexts):
    sc = _quad_scene()
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    rays = make_rays([[0.75, 0.25, -2], [0.25, 0.75, -2],            # 0,1: barycentrics follow the ORIGINAL vertex order (Kernels.h:223-239)
                      [0.75, 0.25, 3],                              # 2: back face (sign-flip path, Kernels.h:60-66,85)
                      [0.75, 0.25, -2],                             # 3: minT exactly at the hit -> rejected (open interval, Kernels.h:89)   [minT set below]
                      [0.75, 0.25, -2],                             # 4: maxT exactly at a flat axis-aligned quad -> the slab test's sentinel culls the box (Kernels.h:131-134,190-194)
                      [0.5, 0.5, -1],                               # 5: exactly on the shared diagonal: a tie inside the pair (Kernels.h:97)
                      [5.25, 5.25, 0], [5.9, 5.9, 0],               # 6,7: the unpaired triangle (p3 = p1 => n2 = 0) and its phantom second half
                      [0.75, 0.25, -2], [0.75, 0.25, -2],           # 8,9: axis-parallel directions (components 0 / -0 -> +-1e-10, Kernels.h:149-157)
                      [0.3, 0.6, -2], [-3, -3, -2]],                # 10: plain hit of triangle 1; 11: misses everything
                     [[0, 0, 1], [0, 0, 1], [0, 0, -1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [-0.0, 0.0, 1], [0, 0, 1], [0, 0, 1]])
    rays["minT"][3] = 2.0
    rays["maxT"][4] = 2.0
    ref = run_everywhere(contexts, blobs, rays, "quad KATs", max_ties=2, arbiter=sc)      # (the compressed 4-wide kernels enter the box ray 4 is culled at and report the hit at t = maxT: arbiter-confirmed)
    assert list(ref["triangle"][[0, 1, 2, 6, 8, 9, 10]]) == [0, 1, 0, 2, 0, 0, 1] and list(ref["triangle"][[3, 4, 7, 11]]) == [MISS] * 4
    assert ref["t"][0] == 2.0 and abs(ref["u"][0] - 0.5) < 1e-6 and abs(ref["v"][0] - 0.25) < 1e-6 and ref["t"][2] == 3.0 and ref["t"][6] == 9.0
    assert ref["triangle"][5] in (0, 1) and ref["t"][5] == 1.0


def test_kat_closed_at_maxT(contexts):
    """The pair test is closed at maxT (T <= absDet * tFar accepts T == absDet * tFar, Kernels.h:88): a slanted triangle hit at exactly t = maxT."""
    v = np.array([[0, 0, 0], [2, 0, 2], [0, 2, 2], [5, 5, 9], [6, 5, 9], [5, 6, 9], [7, 7, 9], [8, 7, 9], [7, 8, 9]], np.float32)
    sc = dict(vertices=np.concatenate([v, np.ones((9, 1), np.float32)], 1), indices=np.arange(9, dtype=np.uint32).reshape(3, 3))
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    rays = make_rays([[0.5, 0.5, -2]] * 4, [[0, 0, 1]] * 4)
    rays["maxT"] = [3.0, 2.999, np.nextafter(np.float32(3.0), np.float32(4.0)), 1e6]
    rays["minT"][3] = np.nextafter(np.float32(3.0), np.float32(0.0))          # just below the hit: accepted
    ref = run_everywhere(contexts, blobs, rays, "closed at maxT", max_ties=0, arbiter=sc)
    assert list(ref["triangle"]) == [0, MISS, 0, 0] and ref["t"][0] == 3.0


def test_kat_edge_codes_rotate_barycentrics(contexts):
    """Every edge code the packer emits (Scene.cpp:132-133: 0..2 for the first triangle, 1..3 for the second — 3 behaves as 0,
    Kernels.h:227-235): the reported (u, v) are in ORIGINAL vertex order, whatever edge the pair was merged along."""
    v = np.array([[0, 0, 0], [2, 0, 0], [0, 2, 0], [2, 2, 0], [7, 7, 5], [8, 7, 5], [7, 8, 5]], np.float32)
    vv = np.concatenate([v, np.ones((len(v), 1), np.float32)], 1)
    rays = make_rays([[0.5, 0.5, -1], [1.5, 1.5, -1], [0.25, 1.5, -1], [1.5, 0.25, -1], [7.2, 7.2, 0]], [[0, 0, 1]] * 5)
    codes = set()
    for rot_a in range(3):
        for rot_b in range(3):
            idx = np.array([np.roll([0, 1, 2], rot_a), np.roll([1, 3, 2], rot_b), [4, 5, 6]], np.uint32)
            blobs = orc.build_scene(vv, idx)
            codes |= set((blobs["remap"] >> 30).tolist())
            ref = run_everywhere(contexts, blobs, rays, "edge codes, rotations %d/%d" % (rot_a, rot_b), max_ties=0, arbiter=dict(vertices=vv, indices=idx))
            tri, t, u, w, _ = orc.brute_closest(vv, idx, rays)
            assert np.array_equal(ref["triangle"], tri)
            np.testing.assert_allclose(ref["u"], u, atol=1e-6)
            np.testing.assert_allclose(ref["v"], w, atol=1e-6)
    assert codes == {0, 1, 2, 3}


def test_kat_leaf_of_126_triangles_from_the_builder(contexts):
    """126 triangles with one common bounding box cannot be separated by the sweep: the builder closes ONE leaf of 126 triangles (a split is
    forced only from 127 on, Bvh2.cpp:462-485), all of them unpaired -> a leaf of 126 pairs with degenerate second triangles, walked by the
    kernels' multi-pair leaf loop; with 127 triangles the forced median split gives leaves of 63 and 64.  Built by the PRODUCT's builder."""
    for k, biggest in ((126, 126), (127, 64)):
        sc = book_scene(k)
        host = ra.HostScene(sc["vertices"], sc["indices"], quality=0)
        blobs = orc.build_scene(sc["vertices"], sc["indices"])
        assert host.nodes.tobytes() == blobs["nodes"].tobytes() and host.pairs.tobytes() == blobs["pairs"].tobytes()
        rays = book_rays(sc, k)
        ref = run_everywhere(contexts, host.blobs(), rays, "book of %d pages" % k, min_leaf=biggest, arbiter=sc)
        assert (ref["triangle"] != MISS).sum() >= len(rays) - 2
        assert_matches_arbiter(ref, sc, rays)


# ------------------------------------------------------------------------------------------------------------------------------- big leaves
@pytest.mark.parametrize("arrangement", ["row", "near_first", "near_last", "coincident"])
@pytest.mark.parametrize("n_pairs", [1, 2, 7, 32, 63, 127])
def test_hand_made_leaves(contexts, n_pairs, arrangement):
    """One leaf of 1 .. 127 pairs (the format's maximum, Scene.cpp:294-312): rays into the first pair, the last, every fifth, none, the
    shared diagonal, from behind; stacks in which the first / the last tested pair is the nearest (tFar shrinks once / n times); n
    coincident pairs (the reference keeps the LAST exact-distance tie of a leaf: the V8 family must agree with the oracle bit for bit)."""
    blobs, geometry = leaf_scene(n_pairs, arrangement)
    rays = leaf_rays(n_pairs, arrangement)
    ref = run_everywhere(contexts, blobs, rays, "leaf of %d pairs, %s" % (n_pairs, arrangement), min_leaf=n_pairs, arbiter=geometry)
    assert_matches_arbiter(ref, geometry, rays)
    first = ref["triangle"][0]
    assert first != MISS and first // 2 == {"row": 0, "near_first": 0, "near_last": n_pairs - 1, "coincident": n_pairs - 1}[arrangement]
    if arrangement == "near_last" and n_pairs > 1:
        assert ref["t"][0] == 11.0 and ref["triangle"][1] // 2 == 0      # from the front the LAST pair tested is the nearest (tFar shrank n times); from behind (ray 1) the first


def test_leaf_of_127_lone_triangles_through_the_packer(contexts):
    """The largest leaf the reference's own packer can emit (Scene.cpp:237-261 over a 127-triangle leaf of the BVH2; no two triangles
    share an edge, so every pair holds one triangle and the degenerate second one): 127 pairs, built by the oracle's packer from a
    hand-made BVH2."""
    n = 127
    v, idx = [], []
    for k in range(n + 3):
        x0, z = 2.5 * (k % 16), 10.0 + 1.5 * (k // 16)
        y0 = 3.0 * (k // 16)
        v += [[x0, y0, z], [x0 + 2, y0, z], [x0, y0 + 2, z]]
        idx.append([3 * k, 3 * k + 1, 3 * k + 2])
    vv = np.concatenate([np.array(v, np.float32), np.ones((len(v), 1), np.float32)], 1)
    idx = np.array(idx, np.uint32)
    nodes = np.zeros(3, orc.BVH2_NODE_DTYPE)
    tri_lo, tri_hi = vv[idx][:, :, :3].min(1), vv[idx][:, :, :3].max(1)
    nodes[0]["kind"], nodes[0]["first"], nodes[0]["last"] = 1, 1, 2
    nodes[0]["bbMin"], nodes[0]["bbMax"] = tri_lo.min(0), tri_hi.max(0)
    nodes[1]["first"], nodes[1]["last"], nodes[1]["bbMin"], nodes[1]["bbMax"] = 0, n, tri_lo[:n].min(0), tri_hi[:n].max(0)
    nodes[2]["first"], nodes[2]["last"], nodes[2]["bbMin"], nodes[2]["bbMax"] = n, n + 3, tri_lo[n:].min(0), tri_hi[n:].max(0)
    blobs = orc.scene_pack(nodes, np.arange(n + 3, dtype=np.uint32), vv, idx)
    assert blobs["pair_count"] == n + 3
    centres = vv[idx][:, :, :3].mean(1)
    o = np.concatenate([centres - [0, 0, 20], centres[::4] + [1.2, 1.2, -20.0], centres[::9] + [0, 0, 20]])       # hits; the phantom second halves (misses); from behind
    d = np.concatenate([np.tile([0, 0, 1.0], (len(centres), 1)), np.tile([0, 0, 1.0], (len(centres[::4]), 1)), np.tile([0, 0, -1.0], (len(centres[::9]), 1))])
    rays = make_rays(o, d)
    ref = run_everywhere(contexts, blobs, rays, "127 lone triangles", min_leaf=127, arbiter=dict(vertices=vv, indices=idx))
    assert_matches_arbiter(ref, dict(vertices=vv, indices=idx), rays)
    assert np.array_equal(ref["triangle"][:n + 3], np.arange(n + 3)) and (ref["triangle"][n + 3:n + 3 + len(centres[::4])] == MISS).all()


# ------------------------------------------------------------------------------------------------ other scene classes, large coordinates, slivers
def test_soup_and_city_scenes(contexts):
    """The two other scene classes (rayaccel_amd/synth.py) at reduced size: unconnected triangle soup with heavy overlap — leaves of up to a
    dozen lone triangles, stacks deeper than the LDS part — and the axis-aligned city (long coplanar walls, areas over seven decades)."""
    for sc, min_leaf in ((synth.soup_synth(triangles=30000, clusters=16), 8), (synth.city_synth(blocks=14), 4)):
        blobs = orc.build_scene(sc["vertices"], sc["indices"])
        prim, _ = synth.primary_rays(sc["camera"], 128, 128)
        hits = orc.traverse(blobs, prim)
        rays = np.concatenate([prim, synth.diffuse_bounce_rays(sc, prim, hits, 20000), synth.random_rays(10007, seed=3, ymax=40.0)])
        ref = run_everywhere(contexts, blobs, rays, sc["name"], env=sc["env"][::8, ::8].copy(), min_leaf=min_leaf, arbiter=sc, max_ties=40, reference_ties=40)
        assert 0.1 < (ref["triangle"] != MISS).mean() < 0.95
        sample = np.arange(0, len(rays), 9)
        assert_matches_arbiter(ref[sample], sc, rays[sample], uv_atol=1e-4)


def test_coordinates_at_1e4_to_1e5(contexts):
    """The small battlefield-synth scene scaled by 1000 and moved to (5e4, 2e4, -7e4): binary32 spacing of 0.004 .. 0.016 on every box plane,
    origin and C = p0 - o.  Bit-exact all the same (one expression tree on both sides); the arbiter confirms the hits."""
    sc = far_scene()
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    prim, _ = synth.primary_rays(sc["camera"], 128, 128)
    rnd = synth.random_rays(6000, seed=9, extent=100.0, ymax=30.0)
    rnd["origin"] = rnd["origin"] * np.float32(1000.0) + np.array([5e4, 2e4, -7e4], np.float32)
    rays = np.concatenate([prim, rnd])
    ref = run_everywhere(contexts, blobs, rays, "coordinates at 1e4..1e5", env=sc["env"][::8, ::8].copy(), arbiter=sc, max_ties=8, reference_ties=8)
    assert (ref["triangle"] != MISS).mean() > 0.3
    assert_matches_arbiter(ref, sc, rays)


def test_sliver_triangles(contexts):
    """3,000 needles (length 5 .. 60, width 1e-6 .. 1e-1): the pair test's determinants cancel to a few bits, so against the double-precision
    arbiter the binary32 algorithm itself is good to 5e-3 in t and 1e-2 in u, v here (the oracle, i.e. the reference's arithmetic, just the
    same) — what is asserted bit for bit is that the HIP engine computes exactly what the restated reference computes."""
    sc = sliver_scene()
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    rays = synth.random_rays(20000, seed=3, extent=50.0, ymax=50.0)
    rays["origin"][:, 1] -= 25
    ref = run_everywhere(contexts, blobs, rays, "slivers", arbiter=sc, max_ties=8, reference_ties=8)
    assert (ref["triangle"] != MISS).sum() > 100
    assert_matches_arbiter(ref, sc, rays, rel=5e-3, uv_atol=1e-2)


def test_needles_and_far_coordinates_on_trees_with_spatial_splits(contexts):
    """The same two scenes through the library's default build (quality 1: a needle 60 long and 1e-6 wide is cut into dozens of references, each
    in a leaf of its own; box planes at 1e5 move in steps of 0.008): every instantiation against the oracle on the same blobs — the V8 family
    bit for bit, the wide kernels up to ties and to the same triangle reached through another of its references — and the arbiter's hits."""
    sc = sliver_scene()
    hs = ra.HostScene(sc["vertices"], sc["indices"], quality=1, split_percent=60)
    assert hs.pair_count > 1.3 * len(sc["indices"].reshape(-1, 3))
    rays = synth.random_rays(20000, seed=3, extent=50.0, ymax=50.0)
    rays["origin"][:, 1] -= 25
    ref = run_everywhere(contexts, hs.blobs(), rays, "slivers, split tree", arbiter=sc, max_ties=8, reference_ties=8)
    assert (ref["triangle"] != MISS).sum() > 100
    assert_matches_arbiter(ref, sc, rays, rel=5e-3, uv_atol=1e-2)
    sc = far_scene()
    hs = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
    prim, _ = synth.primary_rays(sc["camera"], 128, 128)
    ref = run_everywhere(contexts, hs.blobs(), prim, "coordinates at 1e4..1e5, split tree", env=sc["env"][::8, ::8].copy(), arbiter=sc, max_ties=8, reference_ties=8)
    assert_matches_arbiter(ref, sc, prim, uv_atol=1e-4)


def test_paced_issue_does_not_starve_the_chain(contexts):
    """ADVICE (round 5): a caller that issues 1M-ray batches at about the GPU's pace — each one while the chain's kernels are in their drain,
    alive with a few long-ray waves — must not have its batches traced by those few waves.  Same results, and the paced sequence may
    not take more than 2.5 x the back-to-back one."""
    import time
    sc = synth.battlefield_synth(grid=120, boxes=200, quads=800)
    host = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    hits = orc.traverse(host.blobs(), prim, threads=8)
    rays = synth.diffuse_bounce_rays(sc, prim, hits, 1 << 20)
    ref = orc.traverse(host.blobs(), rays, env=None, threads=8)
    with ra.Context(device=0) as ctx:                       # default options: chained from 786,432 rays on
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        d_r = ctx.alloc(rays.nbytes); d_r.upload(rays)
        outs = [ctx.alloc(len(rays) * 16) for _ in range(12)]
        def sequence(pause):
            t0 = time.perf_counter()
            for o in outs:
                ctx.intersect_device(scene, None, d_r.ptr, o.ptr, len(rays), lane=ra.LANE_AUTO)
                if pause:
                    time.sleep(pause)
            ctx.wait(ra.LANE_AUTO)
            return time.perf_counter() - t0
        sequence(0.0)
        back_to_back = min(sequence(0.0) for _ in range(3))
        for pause in (0.00015, 0.0003, 0.0006):
            paced = min(sequence(pause) for _ in range(3))
            for o in outs[::5]:
                assert_bit_exact(o.download(orc.RESULT_DTYPE, len(rays)), ref, "paced issue, %.0f us" % (pause * 1e6))
            gpu_bound = max(back_to_back, 12 * pause)
            print("paced issue %.0f us: %.2f ms for 12 batches (back to back %.2f ms)" % (pause * 1e6, paced * 1e3, back_to_back * 1e3))
            assert paced < 2.5 * gpu_bound + 0.002, "12 batches issued every %.0f us took %.2f ms; back to back %.2f ms" % (pause * 1e6, paced * 1e3, back_to_back * 1e3)
        for o in outs:
            o.free()
        d_r.free(); scene.destroy()
