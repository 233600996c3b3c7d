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
    reported distance — i.e. where the reference's traversal itself misses the true closest hit.  Returns the number of
    records that differ (ties + closer hits)."""
    assert got.dtype == ref.dtype == synth.RESULT_DTYPE
    hit_g, hit_r = got["triangle"] != MISS, ref["triangle"] != MISS
    if arbiter is None:
        assert np.array_equal(hit_g, hit_r), "%s hit/miss differs at %s" % (what, np.nonzero(hit_g != hit_r)[0][:8])
    else:
        assert not (hit_r & ~hit_g).any(), "%s: the oracle hits, the engine misses at %s" % (what, np.nonzero(hit_r & ~hit_g)[0][:8])
    diff = (hit_g != hit_r) | (hit_r & ((got["triangle"] != ref["triangle"]) | (got["t"].view(np.uint32) != ref["t"].view(np.uint32)) |
                                        (got["u"].view(np.uint32) != ref["u"].view(np.uint32)) | (got["v"].view(np.uint32) != ref["v"].view(np.uint32))))
    bad = np.nonzero(diff)[0]
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
