"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from oracle import oracle as orc
from rayaccel_amd import synth

MISS = 0xFFFFFFFF


def make_rays(origins, dirs, min_t=0.0, max_t=1e6):
    o = np.asarray(origins, np.float32).reshape(-1, 3)
    d = np.asarray(dirs, np.float32).reshape(-1, 3)
    rays = np.zeros(len(o), synth.RAY_DTYPE)
    rays["origin"], rays["dir"] = o, d
    rays["minT"], rays["maxT"] = min_t, max_t
    return rays


def assert_bit_exact(got, ref, what=""):
    """Integer/index work is bit-exact; t/u/v are bit-exact too because the HIP kernel evaluates the
    same IEEE expression tree as the oracle.  Miss colours go through acosf (libm vs ocml): 1e-5."""
    assert got.dtype == ref.dtype == synth.RESULT_DTYPE
    bad = np.nonzero(got["triangle"] != ref["triangle"])[0]
    assert len(bad) == 0, "%s primId mismatch at %s: got %s want %s" % (what, bad[:8], got["triangle"][bad[:8]], ref["triangle"][bad[:8]])
    hit = ref["triangle"] != MISS
    for f in ("t", "u", "v"):
        g, r = got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)
        bad = np.nonzero(g != r)[0]
        assert len(bad) == 0, "%s %s not bit-exact at %d hits, e.g. got %r want %r" % (what, f, len(bad), got[f][hit][bad[:4]], ref[f][hit][bad[:4]])
        off = np.nonzero(~np.isclose(got[f][~hit], ref[f][~hit], rtol=1e-5, atol=1e-5))[0]
        if len(off):        # where in the batch: a shading kernel that ran too early leaves a contiguous stretch of parked directions
            idx = np.nonzero(~hit)[0][off]
            unit = np.abs(got["t"][idx] ** 2 + got["u"][idx] ** 2 + got["v"][idx] ** 2 - 1.0) < 1e-3
            chunks = np.unique(idx // 64)
            all_miss = np.nonzero(~hit)[0]
            in_those = np.isin(all_miss // 64, chunks)
            what = "%s [%d miss records off, rays %d..%d, %d of them hold a unit vector (an unshaded direction); they lie in %d of %d 64-ray chunks, which hold %d miss records in all; first chunks %s]" % (
                what, len(idx), idx[0], idx[-1], int(unit.sum()), len(chunks), (len(got) + 63) // 64, int(in_those.sum()), chunks[:12].tolist())
        np.testing.assert_allclose(got[f][~hit], ref[f][~hit], rtol=1e-5, atol=1e-5, err_msg="%s miss colour" % what)


QUANT_VARIANTS = (50, 51, 52, 53)      # ... of which the 64 B compressed format: may also report an arbiter-confirmed closer hit
WIDE_VARIANTS = (45, 46, 47, 48, 49, 50, 51, 52, 53)      # kernel_variant rows that traverse a 4-wide device format (racc_kernel_v9.inc; 50-53: the 64 B compressed one, racc_kernel_v10.inc)


def assert_same_closest_hit(got, ref, what="", max_ties=None, arbiter=None):
    """The 4-wide kernels test every box the reference tests, with the reference's arithmetic, but visit hit children in
    their own order (nearest entry first).  The closest hit is therefore the oracle's, bit for bit, except where two
    primitives are hit at the same distance: SURVEY.md §8(c) accepts either there.  Everything else must be bit-exact.

    The compressed 4-wide kernels (racc_kernel_v10.inc) test boxes that CONTAIN the reference's, so they can also find a hit
    the reference's own box test culled although its pair test accepts it (a ray through a box face within rounding): then
    they report a CLOSER hit than the oracle.  Such a record is accepted only with `arbiter` = dict(vertices, indices, rays)
    and only if the double-precision brute force over ALL triangles (SURVEY.md §8(c)'s arbiter) finds its closest hit at the
    reported distance — i.e. where the reference's traversal itself misses the true closest hit.

    A tree built with spatial splits (scene_build.cpp, TriangleSplitter) holds a triangle in several leaves, each with its own pair record —
    alone in one, paired or at another corner of its pair in the next — and a ray that crosses two of them meets the triangle twice, with
    operands that differ and results that agree to rounding; which of the two a kernel keeps depends on its visiting order.  The SAME
    triangle with t within 2e-6 relative (and an absolute floor of an ulp of the coordinates) and u/v within 1e-4 is therefore the same hit
    (`aliases`, counted apart from the ties: up to 0.5 % of the records).  Returns the number of records that differ (ties + closer hits)."""
    assert got.dtype == ref.dtype == synth.RESULT_DTYPE
    hit_g, hit_r = got["triangle"] != MISS, ref["triangle"] != MISS
    if arbiter is None:
        assert np.array_equal(hit_g, hit_r), "%s hit/miss differs at %s" % (what, np.nonzero(hit_g != hit_r)[0][:8])
    else:
        assert not (hit_r & ~hit_g).any(), "%s: the oracle hits, the engine misses at %s" % (what, np.nonzero(hit_r & ~hit_g)[0][:8])
    diff = (hit_g != hit_r) | (hit_r & ((got["triangle"] != ref["triangle"]) | (got["t"].view(np.uint32) != ref["t"].view(np.uint32)) |
                                        (got["u"].view(np.uint32) != ref["u"].view(np.uint32)) | (got["v"].view(np.uint32) != ref["v"].view(np.uint32))))
    alias = diff & hit_g & hit_r & (got["triangle"] == ref["triangle"]) & np.isclose(got["t"], ref["t"], rtol=2e-6, atol=2e-5) & \
        np.isclose(got["u"], ref["u"], rtol=0, atol=1e-4) & np.isclose(got["v"], ref["v"], rtol=0, atol=1e-4)
    assert alias.sum() <= max(8, len(ref) // 200), "%s: %d records report the oracle's triangle through another of its references" % (what, alias.sum())
    bad = np.nonzero(diff & ~alias)[0]
    if max_ties is None:
        max_ties = max(4, len(ref) // 100000)
    assert len(bad) <= max_ties, "%s: %d records differ from the oracle (allowed: %d), e.g. %s" % (what, len(bad), max_ties, bad[:8])
    for i in bad:
        if hit_r[i] and got["triangle"][i] != ref["triangle"][i] and abs(got["t"][i] - ref["t"][i]) <= 1e-6 * abs(ref["t"][i]):
            continue        # a tie: another primitive at the same distance
        closer = hit_g[i] and (not hit_r[i] or got["t"][i] < ref["t"][i])
        assert closer and arbiter is not None, \
            "%s ray %d: got tri %d t=%r, oracle tri %d t=%r — not an exact-distance tie" % (what, i, got["triangle"][i], got["t"][i], ref["triangle"][i], ref["t"][i])
        tri, t, _, _, _ = orc.brute_closest(arbiter["vertices"], arbiter["indices"], arbiter["rays"][i:i + 1])
        assert tri[0] != MISS and abs(t[0] - got["t"][i]) <= 1e-5 * abs(t[0]), \
            "%s ray %d: got tri %d t=%r closer than the oracle's tri %d t=%r, but the arbiter's closest hit is tri %d at t=%r" % (
                what, i, got["triangle"][i], got["t"][i], ref["triangle"][i], ref["t"][i], tri[0], t[0])
    miss = ~hit_r & ~hit_g
    for f in ("t", "u", "v"):
        np.testing.assert_allclose(got[f][miss], ref[f][miss], rtol=1e-5, atol=1e-5, err_msg="%s miss colour" % what)
    return len(bad)


def assert_same_hits_across_trees(a, b, what="", t_floor=2e-6, uv_atol=1e-4, max_other=None):
    """Two trees over the same triangles, one traversal: same hit/miss but for rays that graze an edge within rounding; the same triangle, or
    another one at the same distance — or, for a handful of rays, a closer one hit ON ITS EDGE (a barycentric within 2e-5 of zero: the
    triangle sits alone or at another corner of its pair in that tree, and its edge test rounds the other way); t within 1e-4 relative above a
    floor of an ulp of the coordinates, u/v within 1e-4.  Returns (hit/miss differences, other-triangle records)."""
    n = len(a)
    dis = int(((a["triangle"] == MISS) != (b["triangle"] == MISS)).sum())
    assert dis <= max(2, n // 100000), "%s: %d hit/miss differences between the trees" % (what, dis)
    both = (a["triangle"] != MISS) & (b["triangle"] != MISS)
    other = both & (a["triangle"] != b["triangle"])
    assert other.sum() <= (max(4, n // 50000) if max_other is None else max_other), "%s: %d primIds differ" % (what, other.sum())
    tie = np.isclose(a["t"], b["t"], rtol=1e-6, atol=0)
    nearer = np.where((a["t"] < b["t"])[:, None], np.stack([a["u"], a["v"]], 1), np.stack([b["u"], b["v"]], 1))
    on_edge = np.minimum(np.minimum(nearer[:, 0], nearer[:, 1]), 1.0 - nearer[:, 0] - nearer[:, 1]) < 2e-5
    assert (tie | on_edge)[other].all(), "%s: another triangle is only acceptable at the same distance (or a closer one hit on its edge): rays %s" % (what, np.nonzero(other & ~(tie | on_edge))[0][:8])
    assert (other & ~tie).sum() <= max(2, n // 250000), "%s: %d edge grazers" % (what, (other & ~tie).sum())
    same = both & ~other
    np.testing.assert_allclose(b["t"][same], a["t"][same], rtol=1e-4, atol=t_floor, err_msg=what)
    np.testing.assert_allclose(b["u"][same], a["u"][same], rtol=1e-4, atol=uv_atol, err_msg=what)
    np.testing.assert_allclose(b["v"][same], a["v"][same], rtol=1e-4, atol=uv_atol, err_msg=what)
    return dis, int(other.sum())


def assert_matches_arbiter(res, scene, rays, rel=1e-4, uv_atol=2e-5):
    """SURVEY.md §8(c) acceptance rule against the double-precision brute force (north_star: primId
    exact, t/u/v within 1e-4 rel).  On a tie (two triangles within 1e-6 rel in t) either id is accepted
    provided re-intersecting the REPORTED triangle in double reproduces the reported t,u,v."""
    v, idx = scene["vertices"], scene["indices"]
    tri, t, u, vv, t2 = orc.brute_closest(v, idx, rays)
    hit_g, hit_b = res["triangle"] != MISS, tri != MISS
    # hit/miss may legitimately differ only when the brute-force hit grazes an edge (|u|,|v|,|w| ~ 0)
    dis = np.nonzero(hit_g != hit_b)[0]
    for i in dis:
        if hit_b[i]:
            assert min(u[i], vv[i], 1 - u[i] - vv[i]) < 1e-5, "ray %d: arbiter hits tri %d at t=%g, engine misses" % (i, tri[i], t[i])
        else:
            ok, tt, uu, v2 = orc.brute_one(v, idx, res["triangle"][i], rays[i])
            raise AssertionError("ray %d: engine hits tri %d, arbiter misses (one-triangle recheck: %s)" % (i, res["triangle"][i], ok))
    both = hit_g & hit_b
    same = both & (res["triangle"] == tri)
    np.testing.assert_allclose(res["t"][same], t[same], rtol=rel, atol=0)
    np.testing.assert_allclose(res["u"][same], u[same], rtol=rel, atol=uv_atol)
    np.testing.assert_allclose(res["v"][same], vv[same], rtol=rel, atol=uv_atol)
    for i in np.nonzero(both & ~same)[0]:
        ok, tt, uu, v2 = orc.brute_one(v, idx, res["triangle"][i], rays[i])
        assert ok, "ray %d: reported tri %d is not hit at all in double precision (arbiter: tri %d)" % (i, res["triangle"][i], tri[i])
        assert abs(tt - t[i]) <= 1e-6 * max(1.0, abs(t[i])) * 10, "ray %d: tri %d at t=%g is not a tie with arbiter tri %d at t=%g" % (i, res["triangle"][i], tt, tri[i], t[i])
        assert abs(res["t"][i] - tt) <= rel * abs(tt) and abs(res["u"][i] - uu) <= 2e-4 and abs(res["v"][i] - v2) <= 2e-4
    return int((both & ~same).sum())


def comb_scene(height=40):
    """Hand-made reference-format blob whose traversal stack must reach `height-1` entries: a
    left-leaning chain of inner nodes whose two child boxes both span the whole ray, the RIGHT (far)
    child being the next chain link and the LEFT a one-pair leaf slightly nearer.  Forces the global
    spill levels above the 16 LDS-resident ones."""
    n = height
    nodes = np.zeros(n, orc.GPU_NODE_DTYPE)
    pairs = np.zeros(n + 1 + 31, orc.PAIR_DTYPE)
    remap = np.zeros(2 * (n + 1), np.uint32)
    for k in range(n + 1):
        # triangle k: big quad half at z = 100 - k (nearer with k), unpaired (p3 = p1)
        z = 100.0 - k
        p0, p1, p2 = np.array([-50, -50, z], np.float32), np.array([50, -50, z], np.float32), np.array([0, 50, z], np.float32)
        pairs[k]["e1"], pairs[k]["e2"], pairs[k]["p0"] = p0 - p1, p2 - p0, p0
        e3 = p1 - p0
        pairs[k]["e3x"], pairs[k]["e3y"], pairs[k]["e3z"] = e3
        remap[2 * k] = k
    for k in range(n):
        nodes[k]["kind"] = 1
        # LEFT child = next chain link (or the last leaf): box z in [0, 200]; entry t is smaller -> descended first
        # RIGHT child = leaf k: box z in [1, 200] -> farther, pushed
        nodes[k]["first"] = (0x80000000 | (k + 1)) if k + 1 < n else ((1 << 24) | n)
        nodes[k]["last"] = (1 << 24) | k
        nodes[k]["leftMin"], nodes[k]["leftMax"] = (-60, -60, 0), (60, 60, 200)
        nodes[k]["rightMin"], nodes[k]["rightMax"] = (-60, -60, 1), (60, 60, 200)
    pairs[n + 1:] = pairs[0]
    npad = n + 1
    while (npad * 3) % 32:
        npad += 1
    return dict(nodes=nodes, pairs=pairs[:npad].copy(), remap=remap, pair_count=n + 1)


# ---------------------------------------------------------------------------------------------------------------- hand-made leaves (round 6)
def _pad_pairs(pairs, count):
    npad = count
    while (npad * 3) % 32:                # Scene.cpp:334-338: copies of pair 0 until 3 * count % 32 == 0
        npad += 1
    out = np.zeros(npad, orc.PAIR_DTYPE)
    out[:count] = pairs[:count]
    out[count:] = pairs[0]
    return out


def quad_pair(a, b, c, d):
    """The 48-byte record of the pair made of triangles (a, c, b) and (a, d, c) — they share the diagonal a-c, the pair's p0-p1 edge
    (Scene.cpp:149-153: e1 = p0 - p1, e2 = p2 - p0, e3 = p3 - p0 with p0 = a, p1 = c, p2 = b, p3 = d)."""
    a, b, c, d = (np.asarray(x, np.float32) for x in (a, b, c, d))
    rec = np.zeros(1, orc.PAIR_DTYPE)[0]
    rec["e1"], rec["e2"], rec["p0"] = a - c, b - a, a
    e3 = d - a
    rec["e3x"], rec["e3y"], rec["e3z"] = e3
    return rec


def leaf_scene(n_pairs, arrangement="row"):
    """Hand-made reference-format blobs whose root's LEFT child is ONE leaf of `n_pairs` triangle pairs (the format allows 127,
    Scene.cpp:294-312; the builder closes leaves of up to 126 triangles, Bvh2.cpp:467-485) and whose right child is a lone far triangle.
    Pair k is the unit quad [x0, x0+1] x [0, 1] at depth z_k, triangle ids 2k (a, c, b) and 2k+1 (a, d, c) with their index lists rotated so
    that the edge codes cycle through every value the packer emits (Scene.cpp:132-133: first 0..2, second 1..3) — truthfully: the arbiter,
    which only sees the triangles, must find the same (u, v).
      "row"         x0 = 3k, z_k = 10 + (7k mod 5): a ray along +z hits at most one pair
      "near_first"  x0 = 0,  z_k = 10 + k/2: every pair is hit, the first tested one is the nearest (tFar shrinks once)
      "near_last"   x0 = 0,  z_k = 10 + (n-1-k)/2: every pair is hit and each one is nearer than the one before (tFar shrinks n times)
      "coincident"  x0 = 0,  z_k = 10: n exact-distance ties — the reference keeps the LAST one tested (T <= absDet * tFar, Kernels.h:88)
    Returns (blobs, geometry) with geometry = dict(vertices, indices) for the arbiter."""
    n = int(n_pairs)
    assert 1 <= n <= 127
    pairs = np.zeros(n + 1, orc.PAIR_DTYPE)
    remap = np.zeros(2 * (n + 1), np.uint32)
    verts, idx = [], []
    lo, hi = np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)
    for k in range(n):
        x0 = 3.0 * k if arrangement == "row" else 0.0
        z = {"row": 10.0 + (7 * k) % 5, "near_first": 10.0 + 0.5 * k, "near_last": 10.0 + 0.5 * (n - 1 - k), "coincident": 10.0}[arrangement]
        a, b, c, d = [x0, 0, z], [x0 + 1, 0, z], [x0 + 1, 1, z], [x0, 1, z]
        pairs[k] = quad_pair(a, b, c, d)
        # the original triangles' index order decides the edge codes (Scene.cpp:127-136): first triangle = (a, c, b) rotated so that a sits at
        # position e0 (its edge e0 is the shared a->c), second = (c, a, d) rotated so that c sits at position e1 (its edge e1 is c->a)
        e0, e1 = k % 3, (k // 3) % 3
        remap[2 * k] = (2 * k) | (e0 << 30)
        remap[2 * k + 1] = (2 * k + 1) | ((e1 + 1) << 30)
        base = len(verts)
        verts += [a, b, c, d]
        idx += [np.roll([base, base + 2, base + 1], e0).tolist(), np.roll([base + 2, base, base + 3], e1).tolist()]
        lo = np.minimum(lo, np.array(a, np.float32)); hi = np.maximum(hi, np.array(c, np.float32))
    # the lone triangle (unpaired: p3 = p1, Scene.cpp:174-178), far off to the side
    p0, p1, p2 = np.array([-20, 0, 30], np.float32), np.array([-18, 0, 30], np.float32), np.array([-20, 2, 30], np.float32)
    pairs[n]["e1"], pairs[n]["e2"], pairs[n]["p0"] = p0 - p1, p2 - p0, p0
    pairs[n]["e3x"], pairs[n]["e3y"], pairs[n]["e3z"] = p1 - p0
    remap[2 * n] = 2 * n
    base = len(verts)
    verts += [p0, p1, p2]
    idx += [[base, base + 1, base + 2]]
    nodes = np.zeros(1, orc.GPU_NODE_DTYPE)
    nodes[0]["kind"] = 1
    nodes[0]["first"] = (n << 24) | 0
    nodes[0]["last"] = (1 << 24) | n
    nodes[0]["leftMin"], nodes[0]["leftMax"] = lo, hi
    nodes[0]["rightMin"], nodes[0]["rightMax"] = (-20, 0, 30), (-18, 2, 30)
    v = np.array(verts, np.float32)
    geometry = dict(vertices=np.concatenate([v, np.ones((len(v), 1), np.float32)], 1), indices=np.array(idx, np.uint32))
    return dict(nodes=nodes, pairs=_pad_pairs(pairs, n + 1), remap=remap, pair_count=n + 1), geometry


def leaf_rays(n_pairs, arrangement):
    """Rays for leaf_scene: into the first pair, the last, every fifth, between pairs / beside the leaf (no pair), onto the shared diagonal
    (a tie inside a pair), the lone triangle, tilted ones that cross several pairs' boxes, and a back-face set from behind."""
    n = int(n_pairs)
    ks = sorted(set([0, n - 1] + list(range(0, n, 5))))
    o, d = [], []
    for k in ks:
        x0 = 3.0 * k if arrangement == "row" else 0.0
        for (px, py) in ((0.75, 0.25), (0.25, 0.75), (0.5, 0.5), (0.999, 0.001)):
            o.append([x0 + px, py, -1.0]); d.append([0, 0, 1])
            o.append([x0 + px, py, 100.0]); d.append([0, 0, -1])                        # from behind: back faces, reversed test order in depth
        o.append([x0 + 1.5, 0.5, -1.0]); d.append([0, 0, 1])                              # between two quads of the row / beside the stack: no pair
        o.append([x0 + 0.5, 0.5, -1.0]); d.append([0.02 * (k % 7 - 3), 0.01, 1])          # tilted
    o.append([-19.5, 0.5, 0.0]); d.append([0, 0, 1])                                      # the lone triangle
    o.append([-18.2, 1.8, 0.0]); d.append([0, 0, 1])                                      # ... its phantom second half: a miss
    o.append([-30.0, 0.5, 12.0]); d.append([1, 0.001, 0.02])                              # along the row, grazing
    dd = np.array(d, np.float64)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    return make_rays(o, dd)


def sliver_scene(count=3000, seed=77):
    """Needle triangles whose area goes to zero: length 5 .. 60, width 1e-6 .. 1e-1 (log-uniform), random orientation, no shared
    vertices, in a 100^3 box — the pair test's determinants lose most of their bits."""
    k = np.arange(count)
    ctr = np.stack([(synth.hash_uniform(k, 1, seed) * 2 - 1) * 50, (synth.hash_uniform(k, 2, seed) * 2 - 1) * 50, (synth.hash_uniform(k, 3, seed) * 2 - 1) * 50], 1).astype(np.float64)
    def unit(s0):
        z = synth.hash_uniform(k, s0, seed).astype(np.float64) * 2 - 1
        ph = synth.hash_uniform(k, s0 + 1, seed).astype(np.float64) * 2 * np.pi
        r = np.sqrt(np.maximum(0.0, 1 - z * z))
        return np.stack([r * np.cos(ph), z, r * np.sin(ph)], 1)
    along, side = unit(4), unit(6)
    side -= along * (side * along).sum(1, keepdims=True)
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    length = (5 + 55 * synth.hash_uniform(k, 8, seed).astype(np.float64))[:, None]
    width = (1e-6 * 1e5 ** synth.hash_uniform(k, 9, seed).astype(np.float64))[:, None]
    v = np.stack([ctr - along * length * 0.5, ctr + along * length * 0.5, ctr + side * width], 1).reshape(-1, 3).astype(np.float32)
    return dict(vertices=np.concatenate([v, np.ones((len(v), 1), np.float32)], 1), indices=np.arange(3 * count, dtype=np.uint32).reshape(-1, 3))


def far_scene(scale=1000.0, shift=(5e4, 2e4, -7e4)):
    """The small battlefield-synth scene blown up and moved so that every coordinate sits at 1e4 .. 2e5 (binary32 spacing 0.004 .. 0.016):
    box planes, origins and the pair test's C = p0 - o all round visibly."""
    sc = synth.battlefield_synth(grid=40, boxes=32, quads=100)
    sh = np.array(shift, np.float32)
    v = sc["vertices"].copy()
    v[:, :3] = v[:, :3] * np.float32(scale) + sh
    cam = dict(sc["camera"])
    cam["origin"] = (np.asarray(cam["origin"], np.float32) * np.float32(scale) + sh).astype(np.float32)
    cam["target"] = (np.asarray(cam["target"], np.float32) * np.float32(scale) + sh).astype(np.float32)
    out = dict(sc, vertices=np.ascontiguousarray(v), camera=cam)
    return out


def compare_with_reference_kernel(ref, other, what, rel=1e-4, max_ties=None):
    """The reference's kernel is built with its own fast-math options (RayAccelerator.cpp:489-490), so it is compared within
    north_star's tolerance: NO hit/miss disagreement is accepted; a different primId only as a tie (both report the same
    distance to 1e-5: coplanar or edge-sharing triangles, the later test wins in one arithmetic and not in the other)."""
    n = len(ref)
    hit_r, hit_o = ref["triangle"] != MISS, other["triangle"] != MISS
    disagreements = int((hit_r != hit_o).sum())
    both = hit_r & hit_o
    diff = both & (ref["triangle"] != other["triangle"])
    print("%s: %d rays, %d hit/miss disagreements, %d primId ties" % (what, n, disagreements, int(diff.sum())))
    assert disagreements == 0, "%s: %d hit/miss disagreements" % (what, disagreements)
    assert np.allclose(ref["t"][diff], other["t"][diff], rtol=1e-5), "%s: %d primId mismatches that are not ties" % (what, diff.sum())
    assert diff.sum() <= (max(2, n // 20000) if max_ties is None else max_ties), "%s: %d ties" % (what, diff.sum())
    same = both & ~diff
    np.testing.assert_allclose(other["t"][same], ref["t"][same], rtol=rel, err_msg=what)
    # (u/v floor: in a tree with spatial splits a triangle has several pair records; two arithmetics can keep different ones — 3.8e-5 on one
    #  of 677,778 hits)
    np.testing.assert_allclose(other["u"][same], ref["u"][same], rtol=rel, atol=5e-5, err_msg=what)
    np.testing.assert_allclose(other["v"][same], ref["v"][same], rtol=rel, atol=5e-5, err_msg=what)
    return int(diff.sum())


def book_scene(k=126):
    """`k` triangles that ALL have the unit cube as their bounding box — the "pages" of a book around the cube's diagonal: (0,0,0), (1,1,1) and
    a third vertex on a circle around the diagonal's midpoint — so no sweep position separates them and the builder closes ONE leaf of k
    triangles (SAH: 2 + k > k; Bvh2.cpp:462-485 forces a split only from 127 on).  No shared vertex indices: k lone pairs.  Plus four far
    triangles so that the root is an inner node."""
    e1 = np.array([1.0, -1.0, 0.0]) / np.sqrt(2.0)
    e2 = np.array([1.0, 1.0, -2.0]) / np.sqrt(6.0)
    v, idx = [], []
    for i in range(k):
        a = 2.0 * np.pi * i / k
        r = np.array([0.5, 0.5, 0.5]) + 0.4 * (np.cos(a) * e1 + np.sin(a) * e2)
        v += [[0, 0, 0], [1, 1, 1], r]
        idx.append([3 * i, 3 * i + 1, 3 * i + 2])
    for j in range(4):
        b = len(v)
        v += [[10 + 3 * j, 0, 0], [11 + 3 * j, 0, 0], [10 + 3 * j, 1, 0]]
        idx.append([b, b + 1, b + 2])
    vv = np.concatenate([np.array(v, np.float32), np.ones((len(v), 1), np.float32)], 1)
    return dict(vertices=vv, indices=np.array(idx, np.uint32))


def book_rays(sc, k=126):
    """At every page's centroid from a point three units out along the page's normal (both sides), plus a fan through the diagonal's midpoint."""
    v = sc["vertices"][:, :3].astype(np.float64)
    tri = v[sc["indices"][:k]]
    c = tri.mean(1)
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    o = np.concatenate([c + 3 * n, c - 3 * n, np.tile([[3.0, -2.0, 0.7]], (40, 1)), [[10.25, 0.25, -1.0], [30.0, 30.0, -1.0]]])
    t = np.linspace(0.05, 0.95, 40)[:, None]
    through = np.array([[0.5, 0.5, 0.5]]) + (t - 0.5) * np.array([[0.3, 0.2, -0.6]])
    d = np.concatenate([-n, n, through - np.array([[3.0, -2.0, 0.7]]), [[0, 0, 1.0], [0, 0, 1.0]]])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return make_rays(o, d)
