"""GPU parity tests proper: the HIP path, called through the C-ABI, against the oracle on the same
seeded inputs.  Bar: primId bit-exact; t/u/v bit-exact as well (same IEEE expression tree on both sides;
north_star only asks for 1e-4 rel); miss colours within 1e-5 (acosf differs between libm and ocml)."""
import os
import threading

import numpy as np
import pytest

import rayaccel_amd as ra
from oracle import oracle as orc
from rayaccel_amd import synth
from rayaccel_amd.shard import shard_range
from helpers import MISS, QUANT_VARIANTS, WIDE_VARIANTS, assert_bit_exact, assert_matches_arbiter, assert_same_closest_hit, comb_scene, make_rays

pytestmark = pytest.mark.gpu


def _batches(small):
    prim = small["primary"]
    hits = orc.traverse(small["blobs"], prim)
    return {"primary": prim, "diffuse": synth.diffuse_bounce_rays(small["sc"], prim, hits, 50000),
            "random": synth.random_rays(30011, seed=11, ymax=30.0)}


@pytest.mark.parametrize("kind", ["primary", "diffuse", "random"])
def test_bit_exact_vs_oracle(gpu_ctx, small, kind):
    rays = _batches(small)[kind]
    ref = orc.traverse(small["blobs"], rays, env=small["sc"]["env"])
    got = gpu_ctx.intersect(small["scene"], small["env"], rays)
    assert_bit_exact(got, ref, kind)


def test_golden_vectors_on_gpu(gpu_ctx):
    import json, os
    g = os.path.join(os.path.dirname(__file__), "golden")
    data = np.load(os.path.join(g, "golden_small.npz"))
    scene = gpu_ctx.upload_scene(data["nodes"].view(orc.GPU_NODE_DTYPE).reshape(-1), data["pairs"].view(orc.PAIR_DTYPE).reshape(-1), data["remap"])
    env = gpu_ctx.create_environment(data["env"])
    got = gpu_ctx.intersect(scene, env, data["rays"].view(orc.RAY_DTYPE).reshape(-1))
    assert_bit_exact(got, data["results"].view(orc.RESULT_DTYPE).reshape(-1), "golden")
    assert int((got["triangle"] != MISS).sum()) == json.load(open(os.path.join(g, "golden_small.json")))["hits"]
    scene.destroy(); env.destroy()


def test_arbiter_acceptance(gpu_ctx, small):
    """north_star's criterion judged by the independent double-precision brute force."""
    rays = np.concatenate([small["primary"][::16], _batches(small)["diffuse"][:3000]])
    got = gpu_ctx.intersect(small["scene"], small["env"], rays)
    assert_matches_arbiter(got, small["sc"], rays)


def test_create_scene_end_to_end(gpu_ctx, small):
    """≙ racc::createScene: product host build with the library's default options (the quality-1 tree since round 6) + upload.  Bit-exact
    against the oracle on the blobs that build produces; against the oracle's own build of the mesh (the reference builder's tree): the same
    closest hit — t/u/v to rounding, because a triangle may sit in another pair."""
    sc = small["sc"]
    scene = gpu_ctx.create_scene(sc["vertices"], sc["indices"])
    host = ra.HostScene(sc["vertices"], sc["indices"], quality=None)
    assert host.quality == 1 and scene.info["max_leaf_pairs"] == 1
    got = gpu_ctx.intersect(scene, small["env"], small["primary"])
    assert_bit_exact(got, orc.traverse(host.blobs(), small["primary"], env=sc["env"]), "createScene")
    ref0 = orc.traverse(orc.build_scene(sc["vertices"], sc["indices"]), small["primary"], env=sc["env"])
    hit = ref0["triangle"] != MISS
    assert np.array_equal(hit, got["triangle"] != MISS)
    same = hit & (got["triangle"] == ref0["triangle"])
    assert same.sum() >= hit.sum() - 4                                     # (the rest: exact-distance ties)
    np.testing.assert_allclose(got["t"][hit], ref0["t"][hit], rtol=1e-5)
    np.testing.assert_allclose(got["u"][same], ref0["u"][same], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(got["v"][same], ref0["v"][same], rtol=1e-4, atol=2e-5)
    scene.destroy()


@pytest.mark.parametrize("count", [0, 1, 63, 64, 65, 257, 4097])
def test_ragged_counts(gpu_ctx, small, count):
    rays = small["primary"][:count]
    got = gpu_ctx.intersect(small["scene"], small["env"], rays)
    assert len(got) == count
    if count:
        assert_bit_exact(got, orc.traverse(small["blobs"], rays, env=small["sc"]["env"]), "count=%d" % count)


def test_no_environment_gives_black_misses(gpu_ctx, small):
    got = gpu_ctx.intersect(small["scene"], None, small["primary"])
    ref = orc.traverse(small["blobs"], small["primary"], env=None)
    assert_bit_exact(got, ref, "no env")
    miss = ref["triangle"] == MISS
    assert miss.any() and (got["t"][miss] == 0).all()


def test_invalid_and_unbounded_rays(gpu_ctx, small):
    rays = small["primary"][:512].copy()
    rays["dir"][5, 1] = np.nan
    rays["origin"][77, 0] = np.inf
    rays["minT"][130] = -np.inf
    rays["maxT"][200] = np.nan
    rays["maxT"][300:400] = np.inf          # valid: no far limit
    rays["dir"][401] = (0.0, -0.0, 1.0)     # clamped to +-1e-10
    got = gpu_ctx.intersect(small["scene"], small["env"], rays)
    ref = orc.traverse(small["blobs"], rays, env=small["sc"]["env"])
    assert_bit_exact(got, ref, "invalid rays")
    for i in (5, 77, 130, 200):
        assert got["triangle"][i] == MISS and got["t"][i] == 0 and got["u"][i] == 0 and got["v"][i] == 0


def test_deep_stack_spills_to_global(gpu_ctx):
    blobs = comb_scene(40)
    scene = gpu_ctx.upload_scene(blobs["nodes"], blobs["pairs"], blobs["remap"])
    assert scene.info["inner_height"] == 40 and scene.info["spill_levels"] == 28      # default kernel: 12 stack entries in LDS
    o = np.stack([np.linspace(-20, 20, 300), np.linspace(-15, 15, 300), np.full(300, -10.0)], 1)
    rays = make_rays(o, [[0, 0, 1]] * 300)
    ref, _, _, depth = orc.traverse(blobs, rays, counters=True)
    assert depth.max() == 40
    assert_bit_exact(gpu_ctx.intersect(scene, None, rays), ref, "comb")
    scene.destroy()
    with ra.Context(device=0, kernel_variant=41) as ctx:      # the 8-entry instantiation: 32 levels in the spill
        scene = ctx.upload_scene(blobs["nodes"], blobs["pairs"], blobs["remap"])
        assert scene.info["spill_levels"] == 32
        assert_bit_exact(ctx.intersect(scene, None, rays), ref, "comb, 8-entry LDS stack")
        scene.destroy()


def test_wide_kernels_on_the_comb(gpu_ctx):
    """The 4-wide kernels on the tall comb: collapsed, it is still 14 wide levels deep with up to three entries each; the
    7-entry instantiation takes the DEEP door and the global spill, the C++ one its own spill path."""
    blobs = comb_scene(40)
    o = np.stack([np.linspace(-20, 20, 300), np.linspace(-15, 15, 300), np.full(300, -10.0)], 1)
    rays = make_rays(o, [[0, 0, 1]] * 300)
    ref = orc.traverse(blobs, rays)
    for variant in (45, 46, 49, 50, 51, 53):
        with ra.Context(device=0, kernel_variant=variant) as ctx:
            scene = ctx.upload_scene(blobs["nodes"], blobs["pairs"], blobs["remap"])
            assert_same_closest_hit(ctx.intersect(scene, None, rays), ref, "comb, variant %d" % variant)
            scene.destroy()


def test_wide_kernels_full_size(full):
    """The 4-wide kernels (racc_kernel_v9.inc) on BASELINE configs[1]/[2] at full size: the oracle's closest hit bit for
    bit, except exact-distance ties (none occur on this scene)."""
    blobs, sc = full["blobs"], full["sc"]
    ref_prim = orc.traverse(blobs, full["primary"], env=sc["env"], threads=8)
    bounce = synth.diffuse_bounce_rays(sc, full["primary"], ref_prim, 1 << 20)
    ref_bounce = orc.traverse(blobs, bounce, env=sc["env"], threads=8)
    for variant in (45, 49, 50, 53):
        with ra.Context(device=0, kernel_variant=variant) as ctx:
            scene = ctx.upload_scene(full["host"].nodes, full["host"].pairs, full["host"].remap)
            env = ctx.create_environment(sc["env"])
            arb = (lambda rays: dict(vertices=sc["vertices"], indices=sc["indices"], rays=rays)) if variant in QUANT_VARIANTS else (lambda rays: None)
            ties = assert_same_closest_hit(ctx.intersect(scene, env, full["primary"]), ref_prim, "1M coherent, variant %d" % variant, arbiter=arb(full["primary"]))
            ties += assert_same_closest_hit(ctx.intersect(scene, env, bounce), ref_bounce, "1M diffuse, variant %d" % variant, arbiter=arb(bounce))
            print("variant %d: %d records in 2M rays differ from the oracle (exact-distance ties%s)" % (variant, ties, ", arbiter-confirmed closer hits" if variant in QUANT_VARIANTS else ""))
            # overlapped (chained, where the kernel has a chained instantiation) launches over the lanes
            d_r = ctx.alloc(bounce.nbytes); d_r.upload(bounce)
            outs = [ctx.alloc(len(bounce) * 16) for _ in range(6)]      # (a result array belongs to its batch until the wait returns)
            for k in range(6):
                ctx.intersect_device(scene, env, d_r.ptr, outs[k].ptr, len(bounce), lane=ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
            for o in outs:
                assert_same_closest_hit(o.download(orc.RESULT_DTYPE, len(bounce)), ref_bounce, "overlapped, variant %d" % variant, arbiter=arb(bounce))
                o.free()
            d_r.free(); scene.destroy(); env.destroy()


def test_wide_below_switches_small_launches(small_scene, small_host, small):
    """wide_below: launches below the threshold take the 4-wide kernel (seen in the launch's LDS footprint), the others the
    context's own; both agree with the oracle."""
    rays = _batches(small)["diffuse"]
    ref = orc.traverse(small["blobs"], rays, env=small_scene["env"])
    with ra.Context(device=0, wide_below=20000) as ctx:
        scene = ctx.upload_scene(small_host.nodes, small_host.pairs, small_host.remap)
        env = ctx.create_environment(small_scene["env"])
        assert_same_closest_hit(ctx.intersect(scene, env, rays[:19999]), ref[:19999], "below the threshold")
        wide_lds = ctx.launch_info(0)["lds_bytes_per_block"]
        assert_bit_exact(ctx.intersect(scene, env, rays), ref, "above the threshold")
        assert ctx.launch_info(0)["lds_bytes_per_block"] < wide_lds       # 128 B per lane of fetch stage against 64 B
        assert_same_closest_hit(ctx.intersect(scene, env, rays[:64]), ref[:64], "below again (spill area re-sized)")
        scene.destroy(); env.destroy()


def test_malformed_blobs_are_rejected(gpu_ctx, small_host):
    nodes = small_host.nodes.copy()
    with pytest.raises(ra.RaccError):                      # cycle: node 1 points back to the root
        bad = nodes.copy(); bad["first"][1] = 0x80000000
        gpu_ctx.upload_scene(bad, small_host.pairs, small_host.remap)
    with pytest.raises(ra.RaccError):                      # child index out of range
        bad = nodes.copy(); bad["last"][0] = 0x80000000 | len(nodes)
        gpu_ctx.upload_scene(bad, small_host.pairs, small_host.remap)
    with pytest.raises(ra.RaccError):                      # leaf range past the pairs
        bad = nodes.copy(); bad["first"][0] = (5 << 24) | (len(small_host.pairs) - 2)
        gpu_ctx.upload_scene(bad, small_host.pairs, small_host.remap)
    with pytest.raises(ra.RaccError):
        gpu_ctx.upload_scene(nodes[:0], small_host.pairs, small_host.remap)


def test_launch_options_do_not_change_results(small_scene, small_host, small):
    rays = _batches(small)["diffuse"]
    ref = orc.traverse(small["blobs"], rays, env=small_scene["env"])
    for opt in (dict(waves_per_simd=1, refill_min=1, leaf_min=1, chunk=1), dict(waves_per_simd=8, refill_min=64, leaf_min=64, chunk=4096),
                dict(waves_per_simd=3, refill_min=20, leaf_min=7, chunk=100), dict(lanes=1, chunk=64),
                *[dict(kernel_variant=v) for v in ra.engine.available_variants()], *([dict(kernel_variant=18, regroup_period=3)] if 18 in ra.engine.available_variants() else []), dict(tail_active=65), dict(coop_same_pct=100), dict(coop_same_pct=101), dict(kernel_variant=41, coop_same_pct=100, refill_min=5, leaf_min=3, inner_reps=7), dict(leaf_step=2), dict(leaf_step=3), dict(leaf_step=2, kernel_variant=41), dict(drain_prefetch=1), dict(drain_prefetch=3), dict(drain_prefetch=3, tail_active=64, thin_reps=3), dict(drain_prefetch=3, kernel_variant=41, coop_same_pct=101),
                dict(kernel_variant=45, leaf_min=1, inner_reps=1, thin_reps=1), dict(kernel_variant=45, leaf_step=3, refill_min=5, chunk=7), dict(kernel_variant=49, leaf_min=3, inner_reps=7, tail_active=65),
                dict(kernel_variant=50, leaf_min=1, inner_reps=1, thin_reps=1), dict(kernel_variant=50, leaf_step=3, refill_min=5, chunk=7), dict(kernel_variant=53, leaf_min=3, inner_reps=7, tail_active=65), dict(kernel_variant=50, coop_same_pct=101), dict(kernel_variant=53, coop_same_pct=100),
                dict(chunk=1 << 30), dict(thin_reps=1, inner_reps=1), dict(thin_reps=3, tail_active=40, inner_reps=2), dict(thin_reps=64, tail_active=64), dict(inner_reps=16)):
        with ra.Context(device=0, **opt) as ctx:
            scene = ctx.upload_scene(small_host.nodes, small_host.pairs, small_host.remap)
            env = ctx.create_environment(small_scene["env"])
            check = assert_same_closest_hit if opt.get("kernel_variant") in WIDE_VARIANTS else assert_bit_exact
            check(ctx.intersect(scene, env, rays), ref, str(opt))
            check(ctx.intersect(scene, env, rays), ref, str(opt) + " relaunch")   # cursor re-armed by the kernel
            scene.destroy(); env.destroy()


def test_concurrent_lanes_and_device_path(gpu_ctx, small):
    """≙ gpuSubmissionThreads: distinct lanes driven from distinct host threads (RayAccelerator.cpp:711-717)."""
    batches = _batches(small)
    names = list(batches)
    refs = {k: orc.traverse(small["blobs"], batches[k], env=small["sc"]["env"]) for k in names}
    out, errs = {}, []

    def work(lane, k):
        try:
            for _ in range(3):
                out[k] = gpu_ctx.intersect(small["scene"], small["env"], batches[k], lane=lane)
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i, k)) for i, k in enumerate(names)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs
    for k in names:
        assert_bit_exact(out[k], refs[k], "lane " + k)
    # device-resident entry point (bench path)
    rays = batches["random"]
    d_r, d_o = gpu_ctx.alloc(rays.nbytes), gpu_ctx.alloc(len(rays) * 16)
    d_r.upload(rays)
    ms = gpu_ctx.intersect_device_timed(small["scene"], small["env"], d_r.ptr, d_o.ptr, len(rays), 4, lane=3)
    assert len(ms) == 4 and all(m > 0 for m in ms)
    assert_bit_exact(d_o.download(ra.RESULT_DTYPE, len(rays)), refs["random"], "device path")
    gpu_ctx.intersect_device(small["scene"], small["env"], d_r.ptr, d_o.ptr, 1000, lane=2); gpu_ctx.wait(2)
    assert_bit_exact(d_o.download(ra.RESULT_DTYPE, 1000), refs["random"][:1000], "device path async")
    d_r.free(); d_o.free()


def test_order_independence_and_split_batches(gpu_ctx, small):
    rays = _batches(small)["diffuse"][:20000]
    whole = gpu_ctx.intersect(small["scene"], small["env"], rays)
    perm = np.random.default_rng(5).permutation(len(rays))
    shuffled = gpu_ctx.intersect(small["scene"], small["env"], rays[perm])
    assert np.array_equal(shuffled.view(np.uint8).reshape(-1, 16), whole.view(np.uint8).reshape(-1, 16)[perm])
    parts = np.concatenate([gpu_ctx.intersect(small["scene"], small["env"], rays[a:b]) for a, b in ((0, 777), (777, 12000), (12000, 20000))])
    assert np.array_equal(parts.view(np.uint8), whole.view(np.uint8))


# ---------------------------------------------------------------- BASELINE.json full sizes
def test_full_size_1M_coherent_and_diffuse(gpu_ctx, full):
    """BASELINE configs[1] and [2] at full size: bit-exact vs the oracle (it finishes in ~1 s per batch)
    plus size-independent properties."""
    blobs, sc = full["host"].blobs(), full["sc"]
    got = gpu_ctx.intersect(full["scene"], full["env"], full["primary"])
    assert_bit_exact(got, orc.traverse(blobs, full["primary"], env=sc["env"], threads=8), "1M coherent")
    bounce = synth.diffuse_bounce_rays(sc, full["primary"], got, 1 << 20)
    got2 = gpu_ctx.intersect(full["scene"], full["env"], bounce)
    assert_bit_exact(got2, orc.traverse(blobs, bounce, env=sc["env"], threads=8), "1M diffuse")
    # idempotence
    assert np.array_equal(gpu_ctx.intersect(full["scene"], full["env"], bounce).view(np.uint8), got2.view(np.uint8))
    # shrinking maxT to just past the reported t must keep the same triangle and the same bits: never a
    # DIFFERENT triangle, never a nearer t.  ("Just past": t = T*rcp(absDet) is rounded, and the slab test's
    # mad(min, invDir, -o*invDir) form (Kernels.h:122-123) cancels badly for flat axis-aligned leaf boxes, so
    # a maxT within a few 1e-6 of t can cull the leaf — the reference's own behaviour, restated exactly.)
    hit = np.nonzero(got2["triangle"] != MISS)[0][:200000]
    clamp = bounce[hit].copy(); clamp["maxT"] = got2["t"][hit] * np.float32(1.001) + np.float32(1e-3)
    again = gpu_ctx.intersect(full["scene"], full["env"], clamp)
    same = again["triangle"] == got2["triangle"][hit]
    assert (same | (again["triangle"] == MISS)).all() and same.mean() > 0.999
    assert np.array_equal(again["t"][same].view(np.uint32), got2["t"][hit][same].view(np.uint32))
    # open at minT: starting each ray just past its hit distance must find something strictly farther (or nothing)
    beyond = bounce[hit].copy(); beyond["minT"] = got2["t"][hit] * np.float32(1.00001)
    far = gpu_ctx.intersect(full["scene"], full["env"], beyond)
    fh = far["triangle"] != MISS
    assert (far["t"][fh] > got2["t"][hit][fh]).all()
    # reversibility: shooting back from each hit point toward the origin must not be blocked before the origin
    P = bounce["origin"][hit] + bounce["dir"][hit] * got2["t"][hit][:, None]
    back = make_rays(P, -bounce["dir"][hit], min_t=1e-3, max_t=1.0)
    back["maxT"] = got2["t"][hit] * np.float32(0.999) - np.float32(2e-3)
    ok = back["maxT"] > back["minT"]
    blocked = gpu_ctx.intersect(full["scene"], None, back[ok])["triangle"] != MISS
    assert blocked.mean() < 1e-3


def test_full_size_4M_batch(gpu_ctx, full):
    """Maximum-size case: a 4M-ray batch (4 x the bench batch) equals four 1M launches."""
    rays = np.concatenate([full["primary"]] * 4)
    rays["origin"][1 << 20:] += np.float32(0.25)
    big = gpu_ctx.intersect(full["scene"], full["env"], rays)
    q = 1 << 20
    for k in range(4):
        part = gpu_ctx.intersect(full["scene"], full["env"], rays[k * q:(k + 1) * q])
        assert np.array_equal(part.view(np.uint8), big[k * q:(k + 1) * q].view(np.uint8))


def test_many_streams_in_one_launch(gpu_ctx, small):
    """≙ what racc::render hands a GPU thread: several ray streams, one launch, results in place per stream."""
    b = _batches(small)
    streams = [b["primary"][:5000], b["diffuse"][:1], b["random"][:0], b["diffuse"][1:20000], b["random"][:777]]
    outs = gpu_ctx.intersect_streams(small["scene"], small["env"], streams, lane=1)
    for rays, got in zip(streams, outs):
        assert len(got) == len(rays)
        if len(rays):
            assert_bit_exact(got, orc.traverse(small["blobs"], rays, env=small["sc"]["env"]), "stream of %d" % len(rays))


def test_sets_of_streams_enqueued_asynchronously_on_two_lanes(gpu_ctx, small):
    """racc_hip_intersect_streams_async: what a submission thread of racc::render would issue if it did not block — sets of ray streams
    (ragged, one empty) on two lanes alternately, page-locked and pageable, waited for at the end; results in place per stream."""
    b = _batches(small)
    pool = np.ascontiguousarray(np.concatenate([b["primary"], b["diffuse"], b["random"]]))
    ref = orc.traverse(small["blobs"], pool, env=small["sc"]["env"])
    cuts = [0, 5000, 5001, 5001, 30000, 47777, 60000, 60064, 90000, len(pool)]
    sets = [(0, 3), (3, 5), (5, 9)]                                  # three launches: streams [0,3), [3,5), [5,9)
    for pinned in (False, True):
        rays = [np.ascontiguousarray(pool[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)]
        outs = [np.zeros(len(r), ra.RESULT_DTYPE) for r in rays]
        tokens = [gpu_ctx.register_host(a) for a in rays + outs if len(a)] if pinned else []
        try:
            for rep in range(3):
                for o in outs:
                    o[:] = 0
                for k, (lo, hi) in enumerate(sets):
                    gpu_ctx.intersect_streams_async(small["scene"], small["env"], rays[lo:hi], outs[lo:hi], lane=k % 2)
                gpu_ctx.wait(0); gpu_ctx.wait(1)
                for i, o in enumerate(outs):
                    if len(o):
                        assert_bit_exact(o, ref[cuts[i]:cuts[i + 1]], "async stream %d (%s)" % (i, "page-locked" if pinned else "pageable"))
        finally:
            gpu_ctx.wait(ra.LANE_AUTO)
            for t in tokens:
                gpu_ctx.unregister_host(t)


def test_scheduling_statistics_variant(small_scene, small_host, small):
    """The debug instantiations count what the scheduler did; every ray must be loaded exactly once and every node
    visit / pair test of the oracle's count must appear as a live lane in some step."""
    rays = _batches(small)["diffuse"]
    ref, nv, npairs, _ = orc.traverse(small["blobs"], rays, counters=True)
    for variant in [v for v in (9, 12, 21, 42, 47, 52) if v in ra.engine.available_variants()]:
        with ra.Context(device=0, kernel_variant=variant) as ctx:
            scene = ctx.upload_scene(small_host.nodes, small_host.pairs, small_host.remap)
            ctx.read_stats()
            (assert_same_closest_hit if variant in WIDE_VARIANTS else assert_bit_exact)(ctx.intersect(scene, None, rays), ref, "stats variant %d" % variant)
            st = ctx.read_stats()
            assert st["rays_loaded"] == len(rays)
            # Same traversal as the reference: V1 counts every live lane of every step exactly.  V2/V3 count at the vote,
            # and a thin wave's second body also serves lanes that changed kind in the first one, so they under-count.
            if variant == 9:
                assert st["inner_lanes"] == int(nv.sum()) and st["leaf_lanes"] == int(npairs.sum())
            elif variant in (47, 52):     # V9 / V10 in C++: a visit is a 4-wide node, about half the oracle's binary visits
                assert 0.3 * int(nv.sum()) <= st["inner_lanes"] <= 0.8 * int(nv.sum()) and st["leaf_lanes"] > 0
            elif variant == 42:     # V8: the inner and leaf steps run inside the assembly block and are not counted
                assert st["waves"] > 0 and st["refill_iters"] >= st["waves"]
                scene.destroy()
                continue
            else:
                assert 0.9 * int(nv.sum()) <= st["inner_lanes"] <= int(nv.sum()) and st["leaf_lanes"] == int(npairs.sum())
            assert st["inner_iters"] * 64 >= st["inner_lanes"] and st["waves"] > 0
            scene.destroy()


def test_pinned_host_block(gpu_ctx, small):
    lib = ra.load_library()
    import ctypes as C
    rays = small["primary"][:4096].copy()
    buf = np.zeros(4096 * 48 + 8192, np.uint8)
    base = (buf.ctypes.data + 4095) & ~4095
    assert lib.racc_hip_register_host(gpu_ctx._h, C.c_void_p(base), 4096 * 48) == 0
    C.memmove(base, rays.ctypes.data, rays.nbytes)
    assert lib.racc_hip_intersect(gpu_ctx._h, small["scene"]._h, small["env"]._h, C.c_void_p(base), C.c_void_p(base + 4096 * 32), 4096, 0) == 0
    got = np.frombuffer((C.c_char * (4096 * 16)).from_address(base + 4096 * 32), ra.RESULT_DTYPE).copy()
    assert lib.racc_hip_unregister_host(gpu_ctx._h, C.c_void_p(base)) == 0
    assert_bit_exact(got, orc.traverse(small["blobs"], rays, env=small["sc"]["env"]), "pinned block")


def test_watchdog_trip_is_reported(small, small_host, monkeypatch):
    """A wave that hits the iteration limit leaves results unwritten: the next synchronising call must say so (the limit is
    2^24 iterations; RACC_MAX_ITERS lowers it for this test), once, and the context stays usable."""
    monkeypatch.setenv("RACC_MAX_ITERS", "2")       # scheduling headers per wave; rays that hit the terrain need dozens
    with ra.Context(device=0) as ctx:
        scene = ctx.upload_scene(small_host.nodes, small_host.pairs, small_host.remap)
        with pytest.raises(ra.RaccError) as e:
            ctx.intersect(scene, None, small["primary"])
        assert e.value.code == -2 and "watchdog" in str(e.value)
        ctx.synchronize()                                         # reported once
        scene.destroy()
    monkeypatch.delenv("RACC_MAX_ITERS")
    with ra.Context(device=0) as ctx:
        scene = ctx.upload_scene(small_host.nodes, small_host.pairs, small_host.remap)
        rays = small["primary"][:5000]
        assert_bit_exact(ctx.intersect(scene, None, rays), orc.traverse(small["blobs"], rays), "after a watchdog trip elsewhere")
        scene.destroy()


def test_sliced_host_path_with_page_locked_streams(gpu_ctx, full):
    """Page-locked host arrays and >= 512k rays take the pipelined path (copies of one slice beside the kernel of another);
    slice boundaries fall inside the ray streams.  Same results as the plain path, stream by stream, bit for bit."""
    lib = ra.load_library()
    prim = full["primary"]
    sizes = (300001, 1, 400000, 77)                              # 700079 rays -> 3 slices of ~233k
    offs = np.cumsum((0,) + sizes)
    streams = [np.ascontiguousarray(prim[offs[i]:offs[i + 1]]) for i in range(len(sizes))]
    plain = gpu_ctx.intersect_streams(full["scene"], full["env"], streams)           # pageable: unsliced
    outs = [np.zeros(n, ra.RESULT_DTYPE) for n in sizes]
    import ctypes as C
    for a in streams + outs:
        assert lib.racc_hip_register_host(gpu_ctx._h, C.c_void_p(a.ctypes.data), a.nbytes) == 0
    n = len(sizes)
    pr = (C.c_void_p * n)(*[a.ctypes.data for a in streams]); po = (C.c_void_p * n)(*[a.ctypes.data for a in outs]); cn = (C.c_uint32 * n)(*sizes)
    assert lib.racc_hip_intersect_streams(gpu_ctx._h, full["scene"]._h, full["env"]._h, n, pr, po, cn, 1) == 0
    for a in streams + outs:
        assert lib.racc_hip_unregister_host(gpu_ctx._h, C.c_void_p(a.ctypes.data)) == 0
    for got, want in zip(outs, plain):
        assert got.tobytes() == want.tobytes()
    ref = orc.traverse(full["host"].blobs(), streams[2][:50000], env=full["sc"]["env"], threads=8)
    assert_bit_exact(outs[2][:50000], ref, "sliced path")



def test_eight_host_batches_back_to_back_over_the_lanes(gpu_ctx, full):
    """Round 4: the host RayStream path pipelined ACROSS batches.  Eight different page-locked batches (ragged sizes, 100k ... 1M rays)
    are enqueued one after the other with racc_hip_intersect_async on rotating lanes — copy-in of batch k+1 beside the kernel of batch
    k beside the copy-out of batch k-1, all copies of the context on one stream per direction — then waited for; twice, the second
    time while the lanes' staging arrays are being recycled with other sizes.  Every record of every batch equals the device-resident
    path's, bit for bit, and a sample of each equals the oracle's."""
    sc, prim = full["sc"], full["primary"]
    hits = gpu_ctx.intersect(full["scene"], full["env"], prim)
    sizes = [1 << 20, 100003, 700079, 1 << 18, 999999, 524288, 65, 800000]
    batches = [np.ascontiguousarray(synth.diffuse_bounce_rays(sc, prim, hits, n, first_sample=3 + k)) for k, n in enumerate(sizes)]
    want = []
    for b in batches:      # the device-resident path, one batch at a time
        d_r = gpu_ctx.alloc(b.nbytes); d_o = gpu_ctx.alloc(len(b) * 16); d_r.upload(b)
        gpu_ctx.intersect_device(full["scene"], full["env"], d_r.ptr, d_o.ptr, len(b), lane=0); gpu_ctx.wait(0)
        want.append(d_o.download(ra.RESULT_DTYPE, len(b))); d_r.free(); d_o.free()
    outs = [np.zeros(len(b), ra.RESULT_DTYPE) for b in batches]
    tokens = [gpu_ctx.register_host(a) for a in batches + outs]
    try:
        for order in (range(8), (5, 0, 7, 2, 6, 1, 4, 3)):
            for o in outs:
                o[:] = 0
            for i, k in enumerate(order):
                gpu_ctx.intersect_async(full["scene"], full["env"], batches[k], outs[k], lane=i % gpu_ctx.lanes)
            gpu_ctx.wait(ra.LANE_AUTO)
            for k in range(8):
                assert outs[k].tobytes() == want[k].tobytes(), "batch %d differs from the device-resident path" % k
        # one lane only: every enqueue first waits for the batch before it (staging reuse) — still complete and in place
        for o in outs:
            o[:] = 0
        for k in range(8):
            gpu_ctx.intersect_async(full["scene"], full["env"], batches[k], outs[k], lane=2)
        gpu_ctx.wait(2)
        for k in range(8):
            assert outs[k].tobytes() == want[k].tobytes(), "batch %d (one lane)" % k
    finally:
        gpu_ctx.wait(ra.LANE_AUTO)
        for t in tokens:
            gpu_ctx.unregister_host(t)
    for k in (0, 2, 6):
        m = min(len(batches[k]), 40000)
        assert_bit_exact(outs[k][:m], orc.traverse(full["host"].blobs(), batches[k][:m], env=sc["env"], threads=8), "pipelined host batch %d" % k)


def test_soak_random_options_sizes_and_lanes():
    """tools/gpu_fuzz.py: random launch options, kernel variants, batch sizes (1 .. 600k) and 1-4 concurrent lanes with
    back-to-back launches of different sizes on each, every result checked against the oracle."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_fuzz.py"), "16", "11"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_lane_auto_rotation(gpu_ctx, small):
    """RACC_HIP_LANE_AUTO: a single-threaded caller issues batch after batch, the engine rotates them over its lanes (their
    launches overlap on the GPU), one wait for all.  Every batch bit-exact."""
    batches = _batches(small)
    names = list(batches) * 3
    outs = []
    for k in names:
        rays = batches[k]
        d_r = gpu_ctx.alloc(rays.nbytes); d_o = gpu_ctx.alloc(len(rays) * 16); d_r.upload(rays)
        gpu_ctx.intersect_device(small["scene"], small["env"], d_r.ptr, d_o.ptr, len(rays), lane=ra.LANE_AUTO)
        outs.append((k, d_r, d_o, len(rays)))
    gpu_ctx.wait(ra.LANE_AUTO)
    refs = {k: orc.traverse(small["blobs"], batches[k], env=small["sc"]["env"]) for k in batches}
    for k, d_r, d_o, n in outs:
        assert_bit_exact(d_o.download(orc.RESULT_DTYPE, n), refs[k], "auto lane " + k)
        d_r.free(); d_o.free()


def test_six_lanes_in_rotation_with_eight_hardware_queues():
    """GPU_MAX_HW_QUEUES >= 8 (read by the HIP runtime when it starts, so: a fresh process) and launches not chained: the engine
    defaults to six lanes, RACC_HIP_LANE_AUTO rotates over all of them with thin grids; twelve batches issued back to back, every
    one bit-exact.  (Chained launches, the default, keep three lanes whatever the queue count.)"""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np
        import rayaccel_amd as ra
        from rayaccel_amd import synth
        from oracle import oracle as orc
        sc = synth.battlefield_synth(grid=40, boxes=32, quads=100)
        host = ra.HostScene(sc["vertices"], sc["indices"])
        prim, _ = synth.primary_rays(sc["camera"], 256, 256)
        ref = orc.traverse(host.blobs(), prim, env=sc["env"])
        with ra.Context(device=0) as ctx:
            assert ctx.auto_lanes == 3, (ctx.lanes, ctx.auto_lanes)
        with ra.Context(device=0, chain_launches=2) as ctx:
            assert (ctx.lanes, ctx.auto_lanes) == (6, 6), (ctx.lanes, ctx.auto_lanes)
            scene = ctx.upload_scene(host.nodes, host.pairs, host.remap); env = ctx.create_environment(sc["env"])
            d_r = ctx.alloc(prim.nbytes); d_r.upload(prim)
            outs = [ctx.alloc(len(prim) * 16) for _ in range(12)]
            for o in outs:
                ctx.intersect_device(scene, env, d_r.ptr, o.ptr, len(prim), lane=ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
            assert ctx.launch_info(5)["waves_per_simd"] == 1
            for o in outs:
                got = o.download(orc.RESULT_DTYPE, len(prim))
                assert np.array_equal(got["triangle"], ref["triangle"]) and np.array_equal(got["t"][ref["triangle"] != 0xFFFFFFFF].view(np.uint32), ref["t"][ref["triangle"] != 0xFFFFFFFF].view(np.uint32))
        with ra.Context(device=0, lanes=2, chain_launches=2) as ctx:
            assert (ctx.lanes, ctx.auto_lanes) == (2, 2)
        print("ok")
    """)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.pop("RACC_AUTO_LANES", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


def test_chained_launches(small_scene, small_host, small):
    """Chained launches (the default for device-resident batches on the engine's streams): waves that run out of rays in one batch
    go on with the next one issued.  Sequences of batches of very different sizes, with invalid rays, on two scenes alternately
    (a different scene breaks the chain), more launches than the descriptor ring holds (it laps), waits in between — every batch
    bit-exact, and identical to the same sequence with chaining off."""
    batches = _batches(small)
    rng = np.random.default_rng(5)
    pool = np.concatenate([batches["primary"], batches["diffuse"], batches["random"]])
    pool = pool[rng.permutation(len(pool))]
    pool["dir"][::997] = np.nan                       # invalid rays: misses with rgb = 0
    ref = orc.traverse(small["blobs"], pool, env=small_scene["env"])
    other = synth.battlefield_synth(grid=24, boxes=8, quads=30)
    other_host = ra.HostScene(other["vertices"], other["indices"])
    ref_other = orc.traverse(other_host.blobs(), pool[:5000], env=small_scene["env"])
    # (chain_min_rays = 1: every batch is chained, whatever its size — the hard case; 0: the default threshold, under which this pool's
    #  batches of <= 130k rays are plain overlapping launches; variant 50: the compressed 4-wide kernel has a chained instantiation too)
    for chain, variant, chain_min in ((0, 0, 1), (0, 0, 0), (2, 0, 0), (0, 50, 1), (0, 60, 1), (3, 62, 1), (0, 63, 1)):      # (60-63: the kernels with an LDS node cache)
        check = assert_same_closest_hit if variant in WIDE_VARIANTS else assert_bit_exact
        with ra.Context(device=0, chain_launches=chain, kernel_variant=variant, chain_min_rays=chain_min) as ctx:
            scene = ctx.upload_scene(small_host.nodes, small_host.pairs, small_host.remap)
            scene2 = ctx.upload_scene(other_host.nodes, other_host.pairs, other_host.remap)
            env = ctx.create_environment(small_scene["env"])
            d_pool = ctx.alloc(pool.nbytes); d_pool.upload(pool)
            issued = []
            for k in range(300):
                n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 20000, int(rng.integers(1, len(pool)))]))
                off = int(rng.integers(0, len(pool) - n + 1))
                d_o = ctx.alloc(n * 16)
                if k % 37 == 36:            # another scene in between: no chain across it
                    m = min(n, 5000)
                    ctx.intersect_device(scene2, env, d_pool.ptr, d_o.ptr, m, lane=ra.LANE_AUTO)
                    issued.append((d_o, 0, m, ref_other))
                else:
                    ctx.intersect_device(scene, env, d_pool.ptr + off * 32, d_o.ptr, n, lane=ra.LANE_AUTO)
                    issued.append((d_o, off, n, ref))
                if k % 53 == 52:
                    ctx.wait(ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
            for i, (d_o, off, n, want) in enumerate(issued):
                check(d_o.download(orc.RESULT_DTYPE, n), want[off:off + n], "chain_launches=%d, chain_min_rays=%d, kernel_variant=%d, launch %d (%d rays)" % (chain, chain_min, variant, i, n))
                d_o.free()
            d_pool.free(); scene.destroy(); scene2.destroy(); env.destroy()


def test_chained_launches_from_three_threads(gpu_ctx, small):
    """Three host threads issue chained batches on one context at once and wait on their own: the chain is shared, every batch
    bit-exact."""
    batches = _batches(small)
    pool = np.concatenate([batches["primary"], batches["diffuse"]])
    ref = orc.traverse(small["blobs"], pool, env=small["sc"]["env"])
    d_pool = gpu_ctx.alloc(pool.nbytes); d_pool.upload(pool)
    errs = []

    def work(seed):
        rng = np.random.default_rng(seed)
        for _ in range(4):
            outs = []
            for _ in range(40):
                n = int(rng.choice([1, 64, 1000, 20000, int(rng.integers(1, len(pool)))]))
                off = int(rng.integers(0, len(pool) - n + 1))
                d_o = gpu_ctx.alloc(n * 16)
                gpu_ctx.intersect_device(small["scene"], small["env"], d_pool.ptr + off * 32, d_o.ptr, n, lane=ra.LANE_AUTO)
                outs.append((d_o, off, n))
            gpu_ctx.wait(ra.LANE_AUTO)
            for d_o, off, n in outs:
                try:
                    assert_bit_exact(d_o.download(orc.RESULT_DTYPE, n), ref[off:off + n], "thread %d" % seed)
                except AssertionError as e:
                    errs.append(str(e)[:200])
                d_o.free()

    ts = [threading.Thread(target=work, args=(s,)) for s in (1, 2, 3)]
    for t in ts: t.start()
    for t in ts: t.join()
    d_pool.free()
    assert not errs, errs[:3]


def test_chained_launches_full_size(gpu_ctx, full):
    """Twelve 1M-ray diffuse batches back to back (chained) and waits on single lanes in between: a lane's wait returns only when
    its batch is complete, whoever traced it."""
    blobs, sc = full["blobs"], full["sc"]
    ref_prim = orc.traverse(blobs, full["primary"], env=sc["env"], threads=8)
    bounce = synth.diffuse_bounce_rays(sc, full["primary"], ref_prim, 1 << 20)
    want = orc.traverse(blobs, bounce, env=sc["env"], threads=8)
    d_r = gpu_ctx.alloc(bounce.nbytes); d_r.upload(bounce)
    outs = [gpu_ctx.alloc(len(bounce) * 16) for _ in range(12)]
    for o in outs:
        gpu_ctx.intersect_device(full["scene"], full["env"], d_r.ptr, o.ptr, len(bounce), lane=ra.LANE_AUTO)
    gpu_ctx.wait((len(outs) - 1) % gpu_ctx.auto_lanes)          # the lane of the LAST batch: complete => every earlier batch of the chain is
    for o in outs:
        assert_bit_exact(o.download(orc.RESULT_DTYPE, len(bounce)), want, "chained 1M batches")
        o.free()
    gpu_ctx.wait(ra.LANE_AUTO)
    d_r.free()


def test_one_lane_from_two_caller_streams(gpu_ctx, small):
    """A lane owns one ray cursor: two launches on the SAME lane from two different caller streams used to race on it
    (ADVICE r1).  The engine now makes the second wait for the first on the device; both must be exact, many times over."""
    s1, s2 = gpu_ctx.create_stream(), gpu_ctx.create_stream()
    batches = _batches(small)
    a, b = batches["diffuse"], batches["random"]
    ref_a = orc.traverse(small["blobs"], a, env=small["sc"]["env"]); ref_b = orc.traverse(small["blobs"], b, env=small["sc"]["env"])
    bufs = []
    for rays in (a, b):
        d_r = gpu_ctx.alloc(rays.nbytes); d_o = gpu_ctx.alloc(len(rays) * 16); d_r.upload(rays)
        bufs.append((d_r, d_o, len(rays)))
    for rnd in range(6):
        gpu_ctx.intersect_device(small["scene"], small["env"], bufs[0][0].ptr, bufs[0][1].ptr, bufs[0][2], lane=1, stream=s1)
        gpu_ctx.intersect_device(small["scene"], small["env"], bufs[1][0].ptr, bufs[1][1].ptr, bufs[1][2], lane=1, stream=s2)
        gpu_ctx.intersect_device(small["scene"], small["env"], bufs[0][0].ptr, bufs[0][1].ptr, bufs[0][2], lane=1)      # and the lane's own stream
        gpu_ctx.stream_synchronize(s1); gpu_ctx.stream_synchronize(s2); gpu_ctx.wait(1)
        assert_bit_exact(bufs[0][1].download(orc.RESULT_DTYPE, bufs[0][2]), ref_a, "two streams, round %d (a)" % rnd)
        assert_bit_exact(bufs[1][1].download(orc.RESULT_DTYPE, bufs[1][2]), ref_b, "two streams, round %d (b)" % rnd)
    for d_r, d_o, _ in bufs:
        d_r.free(); d_o.free()
    gpu_ctx.destroy_stream(s1); gpu_ctx.destroy_stream(s2)


def test_config3_8M_rays_in_8_shards(gpu_ctx, full):
    """BASELINE configs[3] at full size on one GPU: 8,388,608 first-bounce diffuse rays (8 sample sets), once as ONE launch and
    once as the 8 contiguous shards the 8 ranks of a node take (rayaccel_amd.shard.shard_range): the concatenation must be the
    single launch bit for bit, a 1 % sample of it bit-exact against the oracle, and the shards must gather (RCCL all-gather
    entry of the C-ABI, one rank here) into the same buffer."""
    from rayaccel_amd.shard import shard_range
    sc, blobs = full["sc"], full["blobs"]
    hits = gpu_ctx.intersect(full["scene"], full["env"], full["primary"])
    rays = np.concatenate([synth.diffuse_bounce_rays(sc, full["primary"], hits, 1 << 20, first_sample=k) for k in range(8)])
    n = len(rays)
    assert n == 8 << 20
    d_r = gpu_ctx.alloc(rays.nbytes); d_all = gpu_ctx.alloc(n * 16); d_sh = gpu_ctx.alloc(n * 16); d_r.upload(rays)
    gpu_ctx.intersect_device(full["scene"], full["env"], d_r.ptr, d_all.ptr, n)
    gpu_ctx.wait(0)
    whole = d_all.download(orc.RESULT_DTYPE, n)
    for r in range(8):
        b, e = shard_range(n, r, 8)
        gpu_ctx.intersect_device(full["scene"], full["env"], d_r.ptr + b * 32, d_sh.ptr + b * 16, e - b, lane=ra.LANE_AUTO)
    gpu_ctx.wait(ra.LANE_AUTO)
    sharded = d_sh.download(orc.RESULT_DTYPE, n)
    assert sharded.tobytes() == whole.tobytes()
    pick = np.random.default_rng(3).choice(n, n // 100, replace=False)
    assert_bit_exact(whole[pick], orc.traverse(blobs, np.ascontiguousarray(rays[pick]), env=sc["env"], threads=8), "8M, 1 % sample")
    assert 0.5 < (whole["triangle"] != MISS).mean() < 0.8
    # the C-ABI's RCCL entry (world of one rank: the gather is the identity; N ranks run it unchanged)
    comm = ra.Comm(gpu_ctx, ra.Comm.unique_id(), 0, 1)
    d_g = gpu_ctx.alloc(n * 16)
    comm.allgather_results(d_sh.ptr, d_g.ptr, n)
    gpu_ctx.synchronize()
    assert d_g.download(orc.RESULT_DTYPE, n).tobytes() == whole.tobytes()
    comm.destroy()
    for d in (d_r, d_all, d_sh, d_g):
        d.free()


def test_api_misuse_is_reported_not_fatal(gpu_ctx, small):
    """Error paths of the round-2 entries: each returns a status and text, none aborts."""
    lib = ra.load_library()
    with pytest.raises(ra.RaccError) as e:
        gpu_ctx.kernel_times(0)                                  # context created without time_kernels
    assert e.value.code == -1
    with pytest.raises(ra.RaccError):
        ra.Context(device=0, kernel_variant=1000)
    if 22 not in ra.engine.available_variants():
        with pytest.raises(ra.RaccError) as e:
            ra.Context(device=0, kernel_variant=22)              # a retired kernel generation
        assert "not in this build" in str(e.value)
    with pytest.raises(ra.RaccError):
        ra.Comm(gpu_ctx, ra.Comm.unique_id(), 3, 2)              # rank outside the world
    with pytest.raises(ra.RaccError):
        gpu_ctx.intersect_device(small["scene"], small["env"], 0, 0, 64, lane=ra.LANE_AUTO)      # null device pointers
    with pytest.raises(ra.RaccError):
        gpu_ctx.intersect_device(small["scene"], small["env"], 1, 1, 64, lane=17)                 # lane out of range
    rays = small["primary"][:1000]
    assert_bit_exact(gpu_ctx.intersect(small["scene"], small["env"], rays), orc.traverse(small["blobs"], rays, env=small["sc"]["env"]), "after the misuse")


def test_device_group_shards_a_host_batch(full):
    """Multi-device behind the C-ABI (racc_hip_group_*), rehearsed with the entry list [0, 0, 0] on the one GPU of the box: three
    engine contexts, the scene replicated, a ragged 1,000,003-ray batch cut into contiguous shards of whole 64-ray chunks and
    traced concurrently from three host threads.  Bit-exact, in place, in order."""
    rays = np.ascontiguousarray(np.concatenate([full["primary"][:1000000], full["primary"][:3]]))
    grp = ra.Group([0, 0, 0])
    try:
        assert grp.size == 3
        grp.upload(full["host"].nodes, full["host"].pairs, full["host"].remap, full["sc"]["env"])
        got = grp.intersect(rays)
    finally:
        grp.destroy()
    assert_bit_exact(got, orc.traverse(full["blobs"], rays, env=full["sc"]["env"], threads=8), "device group")


def test_device_group_device_resident_shards(full):
    """racc_hip_group_intersect_device: per-member device shards, issued asynchronously from the members' persistent worker
    threads on the members' own streams (lanes rotated, launches chained), several batches back to back, one wait at the end.
    Entry list [0, 0, 0]: three engine contexts on the one GPU of the box.  Every batch of every member bit-exact."""
    blobs, sc = full["blobs"], full["sc"]
    ref_prim = orc.traverse(blobs, full["primary"], env=sc["env"], threads=8)
    bounce = synth.diffuse_bounce_rays(sc, full["primary"], ref_prim, 1 << 20)
    pool = np.concatenate([full["primary"], bounce])
    ref = np.concatenate([ref_prim, orc.traverse(blobs, bounce, env=sc["env"], threads=8)])
    grp = ra.Group([0, 0, 0])
    try:
        grp.upload(full["host"].nodes, full["host"].pairs, full["host"].remap, sc["env"])
        members = [grp.member(i) for i in range(grp.size)]
        rng = np.random.default_rng(21)
        issued, bufs = [], []
        for batch in range(6):
            total = int(rng.integers(200000, len(pool)))
            off = int(rng.integers(0, len(pool) - total + 1))
            cuts = [shard_range(total, i, grp.size) for i in range(grp.size)]
            if batch == 3:
                cuts[1] = (cuts[1][0], cuts[1][0])               # a member that sits a batch out
            d_r, d_o, cnt = [], [], []
            for m, (b, e) in zip(members, cuts):
                n = e - b
                r = m.alloc(max(n, 1) * 32); o = m.alloc(max(n, 1) * 16)
                if n:
                    r.upload(pool[off + b: off + e])
                d_r.append(r.ptr); d_o.append(o.ptr); cnt.append(n); bufs += [r, o]
                issued.append((o, off + b, n))
            grp.intersect_device(d_r, d_o, cnt)
        grp.wait()
        for o, off, n in issued:
            if n:
                assert_bit_exact(o.download(orc.RESULT_DTYPE, n), ref[off:off + n], "group member shard of %d rays" % n)
        for b in bufs:
            b.free()
    finally:
        grp.destroy()
