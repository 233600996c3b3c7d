"""CPU tests of the optional quality mode of the host scene build (racc_host_scene_build_ex, quality 1 / 2;
rayaccel_amd/csrc/scene_build.cpp: TriangleSplitter + LeafSplitter + TreeOptimizer).  The reference has no such mode, so there is nothing to
restate: what is checked is that the blobs are a valid reference-format scene over the same triangles (Scene.cpp:73-87,
237-339), that the oracle's traversal of them (Kernels.h:139-242) finds what the arbiter finds and what it finds in the
reference builder's tree, that they are deterministic for any thread count — and that they do what they are for (fewer node
visits and pair tests per ray)."""
import ctypes as C

import numpy as np
import pytest

import rayaccel_amd as ra
from oracle import oracle as orc
from rayaccel_amd import synth
from helpers import MISS, assert_matches_arbiter, far_scene, sliver_scene


def _geometry_boxes(sc):
    v = sc["vertices"][:, :3]
    tri = v[sc["indices"].reshape(-1, 3)]
    return tri.min(1), tri.max(1)


def check_tree(hs, sc, one_pair_leaves, splits=False):
    """Structural validity of a reference-format blob set against the mesh it was built from.  splits: the tree was built with spatial
    splits — a triangle may sit in several leaves, each with the box of a PART of it: a leaf's box then lies inside its triangles' box, and
    every point of every triangle lies in the box of at least one leaf that holds the triangle (sampled)."""
    T = len(sc["indices"].reshape(-1, 3))
    nodes, remap = hs.nodes, hs.remap
    assert (len(hs.pairs) * 3) % 32 == 0 and len(hs.pairs) > hs.pair_count              # Scene.cpp:334-338
    ids = remap & 0x3FFFFFFF
    second_real = (remap[1::2] >> 30) != 0
    used = np.concatenate([ids[0::2], ids[1::2][second_real]])
    if splits:
        assert np.array_equal(np.unique(used), np.arange(T)), "every triangle at least once"
    else:
        assert np.array_equal(np.sort(used), np.arange(T)), "every triangle exactly once"
    kids = np.stack([nodes["first"], nodes["last"]], 1)
    inner = kids & 0x80000000 != 0
    refs = kids[inner] & 0x7FFFFFFF
    assert np.array_equal(np.sort(refs), np.arange(1, len(nodes))), "every inner node but the root has exactly one parent"
    parent_of = np.zeros(len(nodes), np.int64)
    parent_of[kids[:, 0][inner[:, 0]] & 0x7FFFFFFF] = np.nonzero(inner[:, 0])[0]
    parent_of[kids[:, 1][inner[:, 1]] & 0x7FFFFFFF] = np.nonzero(inner[:, 1])[0]
    assert (parent_of[1:] < np.arange(1, len(nodes))).all(), "parents are numbered before their children (depth-first numbering)"
    leaves = kids[~inner]
    first, cnt = leaves & 0xFFFFFF, leaves >> 24
    order = np.argsort(first)
    assert cnt.min() >= 1 and cnt.max() <= 127
    assert np.array_equal(first[order][1:], (first + cnt)[order][:-1]) and (first + cnt).max() == hs.pair_count
    if one_pair_leaves:
        assert cnt.max() == 1, "quality builds hold one pair per leaf"
    # boxes: a child box is exactly the union of the triangles below it (bottom-up over the depth-first numbering)
    tlo, thi = _geometry_boxes(sc)
    plo = np.full((hs.pair_count, 3), np.inf, np.float32)
    phi = np.full((hs.pair_count, 3), -np.inf, np.float32)
    a = ids[0:2 * hs.pair_count:2]
    plo, phi = np.minimum(plo, tlo[a]), np.maximum(phi, thi[a])
    b = ids[1:2 * hs.pair_count:2]
    sr = second_real[:hs.pair_count]
    plo[sr], phi[sr] = np.minimum(plo[sr], tlo[b[sr]]), np.maximum(phi[sr], thi[b[sr]])
    nlo = np.zeros((len(nodes), 3), np.float32)
    nhi = np.zeros((len(nodes), 3), np.float32)
    leaf_lo = np.zeros((hs.pair_count, 3), np.float32)      # (splits: the box the tree gives the leaf a pair sits in)
    leaf_hi = np.zeros((hs.pair_count, 3), np.float32)
    for i in range(len(nodes) - 1, -1, -1):
        lo2, hi2 = [], []
        for side, (bmin, bmax) in enumerate((("leftMin", "leftMax"), ("rightMin", "rightMax"))):
            ref = int(kids[i, side])
            if ref & 0x80000000:
                lo, hi = nlo[ref & 0x7FFFFFFF], nhi[ref & 0x7FFFFFFF]
                assert np.array_equal(nodes[bmin][i], lo) and np.array_equal(nodes[bmax][i], hi), "node %d side %d: box is not the union of its children's" % (i, side)
            else:
                f, c = ref & 0xFFFFFF, ref >> 24
                lo, hi = plo[f:f + c].min(0), phi[f:f + c].max(0)
                if splits:
                    assert (nodes[bmin][i] >= lo).all() and (nodes[bmax][i] <= hi).all(), "node %d side %d: a leaf's box reaches beyond its triangles" % (i, side)
                    lo, hi = nodes[bmin][i].copy(), nodes[bmax][i].copy()
                    leaf_lo[f:f + c], leaf_hi[f:f + c] = lo, hi
                else:
                    assert np.array_equal(nodes[bmin][i], lo) and np.array_equal(nodes[bmax][i], hi), "node %d side %d: box is not the union of its triangles" % (i, side)
            lo2.append(lo); hi2.append(hi)
        nlo[i], nhi[i] = np.minimum(lo2[0], lo2[1]), np.maximum(hi2[0], hi2[1])
    if splits:
        # coverage: corners, edge midpoints, centroid and random interior points of every triangle, in double precision
        rng = np.random.default_rng(11)
        w = rng.dirichlet([1.0, 1.0, 1.0], 24)
        w = np.concatenate([np.eye(3), [[.5, .5, 0], [0, .5, .5], [.5, 0, .5], [1 / 3, 1 / 3, 1 / 3]], w])
        tri = sc["vertices"][:, :3].astype(np.float64)[sc["indices"].reshape(-1, 3)]
        pts = np.einsum("sw,twk->tsk", w, tri)                                    # [T, S, 3]
        ref_tri = np.concatenate([a, b[sr]])
        ref_lo = np.concatenate([leaf_lo, leaf_lo[sr]]).astype(np.float64)
        ref_hi = np.concatenate([leaf_hi, leaf_hi[sr]]).astype(np.float64)
        eps = 1e-9 * float(np.abs(tri).max() + 1.0)
        inside = ((pts[ref_tri] >= ref_lo[:, None, :] - eps) & (pts[ref_tri] <= ref_hi[:, None, :] + eps)).all(2)      # [refs, S]
        covered = np.zeros((T, len(w)), bool)
        np.logical_or.at(covered, ref_tri, inside)
        assert covered.all(), "%d sample points of %d triangles lie in no leaf box of their triangle" % ((~covered).sum(), (~covered).any(1).sum())


@pytest.fixture(scope="module")
def scene():
    return synth.battlefield_synth(grid=64, boxes=120, quads=700)        # 11k triangles: a dozen subtrees of the first phase


@pytest.fixture(scope="module")
def batches(scene):
    prim, _ = synth.primary_rays(scene["camera"], 128, 128)
    base = orc.build_scene(scene["vertices"], scene["indices"])
    ref = orc.traverse(base, prim)
    diff = synth.diffuse_bounce_rays(scene, prim, ref, 16384)
    rnd = synth.random_rays(8192, 7, extent=100.0, ymax=30.0)
    return dict(primary=prim, diffuse=diff, random=rnd, base=base)


def test_quality_zero_through_the_options_entry_is_the_reference_build(scene):
    a = ra.HostScene(scene["vertices"], scene["indices"])
    b = ra.HostScene(scene["vertices"], scene["indices"], quality=0, threads=3)
    assert a.nodes.tobytes() == b.nodes.tobytes() and a.pairs.tobytes() == b.pairs.tobytes() and a.remap.tobytes() == b.remap.tobytes()


SPLITS = [-1, 0, 40]      # HostScene(split_percent=): none, the library's choice (10 % of the triangle count, or 30 %), a large budget


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("quality", [1, 2])
def test_quality_blobs_are_a_valid_scene(scene, quality, split):
    hs = ra.HostScene(scene["vertices"], scene["indices"], quality=quality, split_percent=split)
    check_tree(hs, scene, one_pair_leaves=True, splits=split >= 0)
    T = len(scene["indices"].reshape(-1, 3))
    extra = (hs.remap[:2 * hs.pair_count:2] & 0x3FFFFFFF).size + ((hs.remap[1:2 * hs.pair_count:2] >> 30) != 0).sum() - T
    assert (extra == 0) if split < 0 else (0 < extra <= T * (split or 30) // 100), extra      # references beyond one per triangle: within the budget


def test_spatial_splits_on_other_scene_classes():
    """Triangle sizes over seven decades, everything axis-aligned (city-synth: the planes fall on the geometry's own), and unconnected random
    triangles (soup-synth: no pairs at all) — valid trees, every point of every triangle covered, the arbiter's hits."""
    for sc in (synth.city_synth(blocks=10, windows=(2, 3)), synth.soup_synth(triangles=6000, clusters=12)):
        hs = ra.HostScene(sc["vertices"], sc["indices"], quality=1, split_percent=30)
        check_tree(hs, sc, one_pair_leaves=True, splits=True)
        prim, _ = synth.primary_rays(sc["camera"], 128, 128)
        hits = orc.traverse(hs.blobs(), prim)
        rays = np.concatenate([prim[::4], synth.diffuse_bounce_rays(sc, prim, hits, 4096)])
        assert_matches_arbiter(orc.traverse(hs.blobs(), rays), sc, rays, uv_atol=1e-4)


def test_reference_build_passes_the_same_validity_check(scene):
    check_tree(ra.HostScene(scene["vertices"], scene["indices"]), scene, one_pair_leaves=False)


@pytest.mark.parametrize("split", [-1, 25])
@pytest.mark.parametrize("quality", [1, 2])
def test_quality_build_is_identical_for_any_thread_count(quality, split):
    sc = synth.battlefield_synth(grid=96, boxes=300, quads=1500)          # ~26k triangles: ~25 subtrees of <= 1024 leaves, two chunks of the splitter
    blobs = [ra.HostScene(sc["vertices"], sc["indices"], quality=quality, threads=t, split_percent=split) for t in (1, 3, 8)]
    for b in blobs[1:]:
        assert b.nodes.tobytes() == blobs[0].nodes.tobytes() and b.pairs.tobytes() == blobs[0].pairs.tobytes() and b.remap.tobytes() == blobs[0].remap.tobytes()


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("quality", [1, 2])
def test_quality_tree_finds_what_the_arbiter_finds(scene, batches, quality, split):
    hs = ra.HostScene(scene["vertices"], scene["indices"], quality=quality, split_percent=split)
    for name in ("primary", "diffuse", "random"):
        rays = batches[name][:4096]
        res = orc.traverse(hs.blobs(), rays)
        # (u/v of one grazing first-bounce ray of this batch are 3.6e-5 off the double-precision values in binary32 — in the
        #  reference builder's tree just the same: the pair test's rounding, not the tree's doing)
        assert_matches_arbiter(res, scene, rays, uv_atol=1e-4)


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("quality", [1, 2])
def test_quality_tree_against_the_reference_builders_tree(scene, batches, quality, split):
    """Same triangles, same pair test, another tree: the closest hit is the same triangle (or another one at the same
    distance), t/u/v agree to rounding — a triangle may sit in another pair, or at another corner of its pair (north_star:
    primId exact, t/u/v within 1e-4 relative)."""
    hs = ra.HostScene(scene["vertices"], scene["indices"], quality=quality, split_percent=split)
    from helpers import assert_same_hits_across_trees
    ties = 0
    for name in ("primary", "diffuse", "random"):
        rays = batches[name]
        a = orc.traverse(batches["base"], rays)
        b = orc.traverse(hs.blobs(), rays)
        ties += assert_same_hits_across_trees(a, b, name, uv_atol=5e-5)[1]
    assert ties <= 8


def test_quality_tree_costs_less(scene, batches):
    """One pair per leaf: fewer pair tests, fewer algorithmic bytes even on a scene too small for the
    re-insertion to find much."""
    base = orc.traverse(batches["base"], batches["diffuse"], counters=True)
    q1 = orc.traverse(ra.HostScene(scene["vertices"], scene["indices"], quality=1, split_percent=-1).blobs(), batches["diffuse"], counters=True)
    q2 = orc.traverse(ra.HostScene(scene["vertices"], scene["indices"], quality=2, split_percent=-1).blobs(), batches["diffuse"], counters=True)
    npairs = [float(r[2].mean()) for r in (base, q1, q2)]
    assert npairs[1] < 0.9 * npairs[0] and npairs[2] < 0.9 * npairs[0], npairs
    bytes_ = [orc.algorithmic_bytes(r[0], r[1], r[2]) for r in (base, q1, q2)]
    assert bytes_[1] < bytes_[0] and bytes_[2] < bytes_[0]


@pytest.mark.parametrize("split", [0, 60])
def test_spatial_splits_of_needles_and_of_coordinates_at_1e5(split):
    """Where the clipping has least to work with: 3,000 needles of length 5 .. 60 and width 1e-6 .. 1e-1 in random orientation (nearly all of
    their boxes is empty: they take the whole budget, up to dozens of cuts each), and the small battlefield scene scaled by 1000 and moved to
    (5e4, 2e4, -7e4), where a binary32 box plane moves in steps of 0.004 .. 0.016.  Valid trees, every sampled point of every triangle inside a
    leaf box of its triangle, and the arbiter's hits (to the tolerances the reference builder's tree is held to on these scenes)."""
    sl = sliver_scene()
    hs = ra.HostScene(sl["vertices"], sl["indices"], quality=1, split_percent=split)
    check_tree(hs, sl, one_pair_leaves=True, splits=True)
    assert hs.pair_count > len(sl["indices"].reshape(-1, 3))
    rays = synth.random_rays(20000, seed=3, extent=50.0, ymax=50.0)
    rays["origin"][:, 1] -= 25
    res = orc.traverse(hs.blobs(), rays)
    assert (res["triangle"] != MISS).sum() > 100
    assert_matches_arbiter(res, sl, rays, rel=5e-3, uv_atol=1e-2)
    base, n0, p0, _ = orc.traverse(orc.build_scene(sl["vertices"], sl["indices"]), rays, counters=True)
    _, n1, p1, _ = orc.traverse(hs.blobs(), rays, counters=True)
    assert 64.0 * n1.mean() + 48.0 * p1.mean() < 0.8 * (64.0 * n0.mean() + 48.0 * p0.mean()), (n0.mean(), p0.mean(), n1.mean(), p1.mean())      # ... and what the splits are for
    far = far_scene()
    hs = ra.HostScene(far["vertices"], far["indices"], quality=1, split_percent=split)
    check_tree(hs, far, one_pair_leaves=True, splits=True)
    prim, _ = synth.primary_rays(far["camera"], 128, 128)
    assert_matches_arbiter(orc.traverse(hs.blobs(), prim), far, prim, uv_atol=1e-4)      # (an ulp of these coordinates is 0.008: u/v to 4.7e-5 with a triangle's other corner as p0)


def test_spatial_splits_pay_where_boxes_overlap():
    """Unconnected random triangles (soup-synth, 60k): every box overlaps dozens of others.  With 10 % more references a first-bounce ray reads
    >= 8 % fewer bytes than in the same quality tree without splits, with 30 % >= 15 % fewer (measured 12 % / 24 %; at a million triangles
    17 % / 31 %) — and a caller who names no budget gets the 30 % here (the SAH estimate of the tree as built drops by more than 7 % with it),
    the 10 % on the battlefield family."""
    sc = synth.soup_synth(triangles=60000, clusters=24)
    prim, _ = synth.primary_rays(sc["camera"], 256, 256)
    trees = {sp: ra.HostScene(sc["vertices"], sc["indices"], quality=1, split_percent=sp) for sp in (-1, 0, 10, 30)}
    rays = synth.diffuse_bounce_rays(sc, prim, orc.traverse(trees[-1].blobs(), prim), 32768)
    res = {sp: orc.traverse(h.blobs(), rays, counters=True) for sp, h in trees.items()}
    cost = {sp: 64.0 * r[1].mean() + 48.0 * r[2].mean() for sp, r in res.items()}
    assert cost[10] < 0.92 * cost[-1] and cost[30] < 0.85 * cost[-1], cost
    assert trees[0].nodes.tobytes() == trees[30].nodes.tobytes() and trees[0].pairs.tobytes() == trees[30].pairs.tobytes()
    from helpers import assert_same_hits_across_trees
    for sp in (10, 30):      # ... and finds the same hits
        assert_same_hits_across_trees(res[-1][0], res[sp][0], "soup, no splits against %d %%" % sp, max_other=8)
    bf = synth.battlefield_synth(grid=96, boxes=300, quads=1500)
    a, b = (ra.HostScene(bf["vertices"], bf["indices"], quality=1, split_percent=sp) for sp in (0, 10))
    assert a.nodes.tobytes() == b.nodes.tobytes() and a.pairs.tobytes() == b.pairs.tobytes() and a.remap.tobytes() == b.remap.tobytes()


def test_quality_tree_of_the_bench_scene_needs_fewer_visits():
    """battlefield-synth at full size, every fourth ray of the 1M first-bounce batch: >= 8 % fewer inner-node visits,
    >= 15 % fewer pair tests than the reference builder's tree (measured: 51.1 -> 45.9 and 3.45 -> 2.69)."""
    sc = synth.battlefield_synth()
    base = ra.HostScene(sc["vertices"], sc["indices"])
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    ref = orc.traverse(base.blobs(), prim, threads=8)
    diff = synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20)[::4].copy()
    q1 = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
    a = orc.traverse(base.blobs(), diff, counters=True)
    b = orc.traverse(q1.blobs(), diff, counters=True)
    assert b[1].mean() < 0.92 * a[1].mean(), (a[1].mean(), b[1].mean())
    assert b[2].mean() < 0.85 * a[2].mean(), (a[2].mean(), b[2].mean())
    from helpers import assert_same_hits_across_trees
    # (u/v: one of the 170,039 hits lies 2.2e-3 from its ray's origin on a quad 0.1 wide — coordinates ~100, an ulp is 7.6e-6 — and the triangle
    #  has another corner of its pair as p0 in this tree: 3.8e-4 apart; the arbiter tests above hold u/v to 1e-4 on the small scene)
    assert_same_hits_across_trees(a[0], b[0], "bench scene, quality 0 against 1", max_other=8, uv_atol=1e-3)


def test_tiny_and_degenerate_inputs():
    """3 triangles (the smallest scene the format holds), a scene of unconnected triangles (no pair merges at all), and a
    leaf-sized cluster of coincident triangles (zero-area boxes: the forced-median path, Bvh2.cpp:467-485; no plane cuts a box of no
    extent) — without spatial splits, with the default budget and with a large one."""
    for split in (-1, 0, 200):
        v = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 1, 0, 1], [5, 0, 0, 1], [6, 0, 0, 1], [5, 1, 0, 1], [0, 0, 9, 1], [1, 0, 9, 1], [0, 1, 9, 1]], np.float32)
        idx = np.arange(9, dtype=np.uint32)
        sc = dict(vertices=v, indices=idx.reshape(-1, 3))
        for q in (1, 2):
            check_tree(ra.HostScene(v, idx, quality=q, split_percent=split), sc, one_pair_leaves=True, splits=split >= 0)
        rng = np.random.default_rng(5)
        v = np.concatenate([rng.uniform(-10, 10, (300, 3)).astype(np.float32), np.ones((300, 1), np.float32)], 1)
        idx = np.arange(300, dtype=np.uint32)
        sc = dict(vertices=v, indices=idx.reshape(-1, 3))
        hs = ra.HostScene(v, idx, quality=1, split_percent=split)
        check_tree(hs, sc, one_pair_leaves=True, splits=split >= 0)
        assert hs.pair_count == 100 if split < 0 else 100 < hs.pair_count <= 100 + (split or 30)      # (every reference its own single-triangle pair)
        v = np.tile(np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 1, 0, 1]], np.float32), (40, 1))
        idx = np.arange(120, dtype=np.uint32)
        sc = dict(vertices=v, indices=idx.reshape(-1, 3))
        hs = ra.HostScene(v, idx, quality=1, split_percent=split)
        check_tree(hs, sc, one_pair_leaves=True, splits=split >= 0)


def test_callers_without_options_get_the_quality_tree_and_the_environment_can_switch_them(scene, monkeypatch):
    """racc_host_scene_build (no options: racc::createScene, the path-tracing consumers) builds RACC_HOST_BUILD_DEFAULT_QUALITY = 1 — the tree
    bench.py's `value` is measured on — unless RACC_BUILD_QUALITY says otherwise; explicit options always win."""
    monkeypatch.delenv("RACC_BUILD_QUALITY", raising=False)
    q0 = ra.HostScene(scene["vertices"], scene["indices"], quality=0)
    q1 = ra.HostScene(scene["vertices"], scene["indices"], quality=1)
    same = lambda a, b: a.nodes.tobytes() == b.nodes.tobytes() and a.pairs.tobytes() == b.pairs.tobytes() and a.remap.tobytes() == b.remap.tobytes()
    assert not same(q0, q1)
    default = ra.HostScene(scene["vertices"], scene["indices"], quality=None)
    assert default.quality == ra.engine.LIBRARY_DEFAULT_QUALITY == 1 and same(default, q1)
    monkeypatch.setenv("RACC_BUILD_QUALITY", "0")
    via_env = ra.HostScene(scene["vertices"], scene["indices"], quality=None)
    assert via_env.quality == 0 and same(via_env, q0)
    monkeypatch.setenv("RACC_BUILD_QUALITY", "1")
    assert same(ra.HostScene(scene["vertices"], scene["indices"], quality=0, threads=2), q0)      # explicit options win over the environment
    assert same(ra.HostScene(scene["vertices"], scene["indices"]), q0)                            # ... and the harness's integer default is explicit
    import re, os
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "racc_hip.h")).read()
    assert int(re.search(r"#define RACC_HOST_BUILD_DEFAULT_QUALITY (\d+)u", header).group(1)) == ra.engine.LIBRARY_DEFAULT_QUALITY


def test_the_split_budget_can_be_set_from_outside(scene, monkeypatch):
    """RACC_BUILD_SPLIT_PERCENT stands in for a budget the caller did not name (options.split_percent = 0, or no options at all): 0 = no
    splits, n = that percentage, and then the library does not pick the larger budget on its own; a budget in the options wins."""
    same = lambda a, b: a.nodes.tobytes() == b.nodes.tobytes() and a.pairs.tobytes() == b.pairs.tobytes() and a.remap.tobytes() == b.remap.tobytes()
    monkeypatch.delenv("RACC_BUILD_SPLIT_PERCENT", raising=False)
    monkeypatch.delenv("RACC_BUILD_QUALITY", raising=False)
    none, p25 = (ra.HostScene(scene["vertices"], scene["indices"], quality=1, split_percent=sp) for sp in (-1, 25))
    assert not same(none, p25)
    monkeypatch.setenv("RACC_BUILD_SPLIT_PERCENT", "0")
    assert same(ra.HostScene(scene["vertices"], scene["indices"], quality=1), none) and same(ra.HostScene(scene["vertices"], scene["indices"], quality=None), none)
    assert same(ra.HostScene(scene["vertices"], scene["indices"], quality=1, split_percent=25), p25)
    monkeypatch.setenv("RACC_BUILD_SPLIT_PERCENT", "25")
    assert same(ra.HostScene(scene["vertices"], scene["indices"], quality=1), p25) and same(ra.HostScene(scene["vertices"], scene["indices"], quality=None), p25)
    assert same(ra.HostScene(scene["vertices"], scene["indices"], quality=1, split_percent=-1), none)
    q0 = ra.HostScene(scene["vertices"], scene["indices"], quality=0)
    monkeypatch.setenv("RACC_BUILD_SPLIT_PERCENT", "300")
    assert same(ra.HostScene(scene["vertices"], scene["indices"], quality=0), q0)      # the reference's builder has no splits


def test_options_are_validated(scene):
    lib = ra.load_library()
    v = ra.engine._as_verts4(scene["vertices"])
    idx = np.ascontiguousarray(scene["indices"], np.uint32).reshape(-1)
    h = C.c_void_p()
    opt = ra.engine.HostBuildOptions(struct_size=0, quality=1)
    assert lib.racc_host_scene_build_ex(ra.engine._ptr(v), len(v), ra.engine._ptr(idx), idx.size, C.byref(opt), C.byref(h)) == -1 and not h
    assert b"struct_size" in lib.racc_hip_last_error()
    opt = ra.engine.HostBuildOptions(struct_size=C.sizeof(ra.engine.HostBuildOptions), quality=3)
    assert lib.racc_host_scene_build_ex(ra.engine._ptr(v), len(v), ra.engine._ptr(idx), idx.size, C.byref(opt), C.byref(h)) == -1 and not h
    assert b"quality" in lib.racc_hip_last_error()
    opt = ra.engine.HostBuildOptions(struct_size=C.sizeof(ra.engine.HostBuildOptions), quality=1, split_percent=1001)
    assert lib.racc_host_scene_build_ex(ra.engine._ptr(v), len(v), ra.engine._ptr(idx), idx.size, C.byref(opt), C.byref(h)) == -1 and not h
    assert b"split_percent" in lib.racc_hip_last_error()
    opt = ra.engine.HostBuildOptions(struct_size=C.sizeof(ra.engine.HostBuildOptions), quality=0, split_percent=500)      # (quality 0 has no splits: ignored)
    assert lib.racc_host_scene_build_ex(ra.engine._ptr(v), len(v), ra.engine._ptr(idx), idx.size, C.byref(opt), C.byref(h)) == 0 and h
    lib.racc_host_scene_free(h); h = C.c_void_p()
    # a caller compiled against a shorter struct (only struct_size + quality): the rest reads as 0
    opt = ra.engine.HostBuildOptions(struct_size=8, quality=1, threads=77)
    assert lib.racc_host_scene_build_ex(ra.engine._ptr(v), len(v), ra.engine._ptr(idx), idx.size, C.byref(opt), C.byref(h)) == 0 and h
    lib.racc_host_scene_free(h)
