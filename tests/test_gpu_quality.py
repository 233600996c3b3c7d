"""GPU parity on the blobs of the optional quality build (racc_host_scene_build_ex, quality 1 / 2): the same reference-format
scene (Scene.cpp:73-87), fewer node visits per ray.  The bar is the suite's: every record the HIP engine returns is
bit-identical to what the oracle's traversal (Kernels.h:139-242 restated) returns on THE SAME blobs; the reference's own
OpenCL kernel (oracle/_ref) consumes the same blobs and agrees within north_star's tolerance; and the quality tree's hits
are the reference builder's tree's hits (same triangle or an exact-distance tie, t/u/v to rounding)."""
import numpy as np
import pytest

import rayaccel_amd as ra
from oracle import oracle as orc, ref_kernel
from rayaccel_amd import synth
from helpers import MISS, assert_bit_exact

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_q1(gpu_ctx, full):
    """battlefield-synth at full size built with quality 1, next to the session's quality-0 scene."""
    sc = full["sc"]
    host = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
    scene = gpu_ctx.upload_scene(host.nodes, host.pairs, host.remap)
    yield dict(host=host, blobs=host.blobs(), scene=scene)
    scene.destroy()


def _cross_tree(a, b, what):
    """Two trees over the same triangles: tests/helpers.py::assert_same_hits_across_trees."""
    from helpers import assert_same_hits_across_trees
    return assert_same_hits_across_trees(a, b, what, uv_atol=1e-3)      # (u/v of a hit a few 1e-3 from the ray's origin: see tests/test_quality_build.py)


@pytest.mark.parametrize("quality", [1, 2])
def test_small_scene_quality_blobs(gpu_ctx, small_scene, quality):
    host = ra.HostScene(small_scene["vertices"], small_scene["indices"], quality=quality)
    prim, _ = synth.primary_rays(small_scene["camera"], 256, 256)
    blobs = host.blobs()
    rays = np.concatenate([prim, synth.diffuse_bounce_rays(small_scene, prim, orc.traverse(blobs, prim), 40001), synth.random_rays(20003, seed=5, ymax=30.0)])
    scene = gpu_ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = gpu_ctx.create_environment(small_scene["env"])
    try:
        ref = orc.traverse(blobs, rays, env=small_scene["env"])
        assert_bit_exact(gpu_ctx.intersect(scene, env, rays), ref, "quality %d, host batch" % quality)
        for n in (1, 63, 64, 65, 4097):
            assert_bit_exact(gpu_ctx.intersect(scene, env, rays[:n]), ref[:n], "quality %d, %d rays" % (quality, n))
        outs = gpu_ctx.intersect_streams(scene, env, [rays[:5000], rays[5000:5001], rays[5001:5001], rays[5001:70000]], lane=1)
        assert_bit_exact(np.concatenate(outs), ref[:70000], "quality %d, ray streams" % quality)
    finally:
        scene.destroy(); env.destroy()


def test_full_size_1M_coherent_and_diffuse_on_the_quality_tree(gpu_ctx, full, full_q1):
    """BASELINE configs[1] and [2] at full size on the quality-1 blobs: bit-exact against the oracle on the same blobs; the bench's
    headline runs on this tree."""
    sc, blobs = full["sc"], full_q1["blobs"]
    got = gpu_ctx.intersect(full_q1["scene"], full["env"], full["primary"])
    assert_bit_exact(got, orc.traverse(blobs, full["primary"], env=sc["env"], threads=8), "1M coherent, quality 1")
    bounce = synth.diffuse_bounce_rays(sc, full["primary"], got, 1 << 20)
    got2 = gpu_ctx.intersect(full_q1["scene"], full["env"], bounce)
    ref2, nv, npairs, depth = orc.traverse(blobs, bounce, env=sc["env"], counters=True, threads=8)
    assert_bit_exact(got2, ref2, "1M diffuse, quality 1")
    # ... and against the reference builder's tree: the same hits
    base0 = gpu_ctx.intersect(full["scene"], full["env"], full["primary"])
    base2 = gpu_ctx.intersect(full["scene"], full["env"], bounce)
    print("coherent: hit/miss differences, ties:", _cross_tree(base0, got, "1M coherent"))
    print("diffuse: hit/miss differences, ties:", _cross_tree(base2, got2, "1M diffuse"))
    # what the mode is for
    ref0, nv0, np0, _ = orc.traverse(full["blobs"], bounce, counters=True, threads=8)
    assert nv.mean() < 0.92 * nv0.mean() and npairs.mean() < 0.85 * np0.mean(), (nv.mean(), nv0.mean(), npairs.mean(), np0.mean())
    assert int(depth.max()) <= 40


def test_chained_device_batches_on_the_quality_tree(full, full_q1):
    """The bench's own loop on a default-options context (batches of >= 786,432 rays are chained): 24 device-resident 1M-ray batches,
    eight sample sets in rotation, issued back to back over the lanes, every record of every batch against the oracle.
    (Device arrays through the engine's own allocator: importing torch into this process would load a second HIP runtime.)"""
    sc, host = full["sc"], full_q1["host"]
    n = 1 << 20
    with ra.Context(device=0) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        hits = ctx.intersect(scene, env, full["primary"])
        sets = synth.diffuse_bounce_batches(sc, full["primary"], hits, n, range(8))
        refs = [orc.traverse(full_q1["blobs"], r, env=sc["env"], threads=8) for r in sets]
        d_sets = []
        for r in sets:
            d = ctx.alloc(r.nbytes); d.upload(r); d_sets.append(d)
        outs = [ctx.alloc(n * 16) for _ in range(24)]
        for k in range(24):
            ctx.intersect_device(scene, env, d_sets[k % 8].ptr, outs[k].ptr, n, lane=ra.LANE_AUTO)
        ctx.wait(ra.LANE_AUTO)
        for k in range(24):
            assert_bit_exact(outs[k].download(ra.RESULT_DTYPE, n), refs[k % 8], "chained batch %d (sample set %d), quality 1" % (k, k % 8))
        for d in d_sets + outs:
            d.free()
        scene.destroy(); env.destroy()


@pytest.mark.skipif(not ref_kernel.built(), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_reference_kernel_takes_the_quality_blobs(gpu_ctx, full, full_q1):
    """The reference's own OpenCL kernel on the quality-1 blobs and the 1M diffuse batch: no hit/miss disagreement with the oracle or
    with the HIP engine, primIds equal up to exact-distance ties, t/u/v within 1e-4 (it is a fast-math build: not bit-comparable)."""
    from test_gpu_reference_kernel import _compare
    sc, blobs = full["sc"], full_q1["blobs"]
    hits = orc.traverse(blobs, full["primary"], threads=8)
    bounce = synth.diffuse_bounce_rays(sc, full["primary"], hits, 1 << 20)
    reference = ref_kernel.run(blobs, bounce, sc["env"])
    ties = _compare(reference, orc.traverse(blobs, bounce, threads=8), "oracle vs reference kernel, 1M diffuse, quality-1 blobs")
    _compare(reference, gpu_ctx.intersect(full_q1["scene"], None, bounce), "HIP engine vs reference kernel, 1M diffuse, quality-1 blobs")
    assert ties <= 4


def test_wide_and_compressed_kernels_on_the_quality_tree(full, full_q1):
    """The optional 4-wide kernels collapse whatever binary tree they are given (collapseWide at upload): same allowances as on the
    reference builder's tree (tests/helpers.py::assert_same_closest_hit)."""
    from helpers import assert_same_closest_hit
    sc, host = full["sc"], full_q1["host"]
    rays = full["primary"][::4].copy()
    ref = orc.traverse(full_q1["blobs"], rays, env=sc["env"], threads=8)
    for variant in (45, 50):
        with ra.Context(device=0, kernel_variant=variant) as ctx:
            scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
            env = ctx.create_environment(sc["env"])
            got = ctx.intersect(scene, env, rays)
            arb = dict(vertices=sc["vertices"], indices=sc["indices"], rays=rays) if variant == 50 else None
            assert_same_closest_hit(got, ref, "variant %d on quality-1 blobs" % variant, arbiter=arb)
            scene.destroy(); env.destroy()


def test_default_threshold_interleaves_chained_and_stand_alone_batches(full, full_q1):
    """the suite's shared context chains every batch (chain_min_rays = 1), so the shipping default — batches of >= 786,432
    rays chained, smaller ones stand-alone on the SAME lanes, sharing the lane's cursor and spill with the chain ring — was hardly run.
    A default-options context, 1M-ray batches alternating with 64 ... 99,999-ray ones over RACC_HIP_LANE_AUTO, waits on single lanes and
    on all of them in between, 280 chained launches in all (the 256-entry descriptor ring laps), every record of every batch compared."""
    sc, host = full["sc"], full_q1["host"]
    n = 1 << 20
    rng = np.random.default_rng(11)
    with ra.Context(device=0) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        hits = ctx.intersect(scene, env, full["primary"])
        sets = synth.diffuse_bounce_batches(sc, full["primary"], hits, n, range(3)) + [full["primary"]]
        refs = [orc.traverse(full_q1["blobs"], r, env=sc["env"], threads=8) for r in sets]
        d_sets = []
        for r in sets:
            d = ctx.alloc(r.nbytes); d.upload(r); d_sets.append(d)
        big_outs = [ctx.alloc(n * 16) for _ in range(40)]
        small_outs = [ctx.alloc(100000 * 16) for _ in range(40)]
        chained = 0
        for rnd in range(7):
            issued = []
            for k in range(40):
                s = int(rng.integers(0, len(sets)))
                ctx.intersect_device(scene, env, d_sets[s].ptr, big_outs[k].ptr, n, lane=ra.LANE_AUTO)
                issued.append((big_outs[k], s, 0, n))
                chained += 1
                if k % 3 != 2:            # one or two small stand-alone batches behind it, on the lanes next in the rotation
                    for _ in range(1 + k % 2):
                        m = int(rng.choice([64, 1000, 27648, 99999]))
                        off = int(rng.integers(0, n - m)) // 64 * 64
                        slot = len([i for i in issued if i[0] in small_outs])
                        if slot >= len(small_outs):
                            break
                        ctx.intersect_device(scene, env, d_sets[s].ptr + off * 32, small_outs[slot].ptr, m, lane=ra.LANE_AUTO)
                        issued.append((small_outs[slot], s, off, m))
                if k == 17:
                    ctx.wait(1)           # a single lane in between: its batches are complete, the others stay in flight
            ctx.wait(ra.LANE_AUTO)
            for d_o, s, off, m in issued:
                assert_bit_exact(d_o.download(ra.RESULT_DTYPE, m), refs[s][off:off + m], "round %d, set %d, %d rays at %d" % (rnd, s, m, off))
        assert chained > 256
        for d in d_sets + big_outs + small_outs:
            d.free()
        scene.destroy(); env.destroy()


def test_a_caller_who_waits_for_every_batch_gets_stand_alone_launches(full, full_q1):
    """Round 5: starting and finishing a chain costs a third of a lone 1M-ray batch's time (0.46 against 0.35 ms).  On a default-options
    context, after two waits in a row that found a chain of ONE batch, a batch issued while nothing else is in flight is launched stand-alone
    (the lazy chain's kernel has a miss queue in LDS: racc_hip_get_launch_info tells the two apart); the first batch issued while another is
    still in flight starts a chain again.  Every record of every batch against the oracle, across the switches."""
    sc, host = full["sc"], full_q1["host"]
    n = 1 << 20
    with ra.Context(device=0, lanes=3) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        hits = ctx.intersect(scene, env, full["primary"])
        sets = synth.diffuse_bounce_batches(sc, full["primary"], hits, n, range(2))
        refs = [orc.traverse(full_q1["blobs"], r, env=sc["env"], threads=8) for r in sets]
        d_sets = []
        for r in sets:
            d = ctx.alloc(r.nbytes); d.upload(r); d_sets.append(d)
        outs = [ctx.alloc(n * 16) for _ in range(6)]
        lds = []
        for k in range(5):                       # issue, wait, issue, wait ...
            ctx.intersect_device(scene, env, d_sets[k % 2].ptr, outs[0].ptr, n, lane=0)
            ctx.wait(0)
            lds.append(ctx.launch_info(0)["lds_bytes_per_block"])
            assert_bit_exact(outs[0].download(ra.RESULT_DTYPE, n), refs[k % 2], "lone batch %d" % k)
        assert lds[0] == lds[1] and lds[2] == lds[3] == lds[4] and lds[2] < lds[0], lds          # two chained (with the miss queue), then stand-alone
        for rnd in range(2):                     # back to back: the second batch finds the first in flight and starts a chain; the rest follow it
            for k in range(6):
                ctx.intersect_device(scene, env, d_sets[k % 2].ptr, outs[k].ptr, n, lane=k % 3)
            assert ctx.launch_info(1)["lds_bytes_per_block"] == lds[0]
            ctx.wait(ra.LANE_AUTO)
            for k in range(6):
                assert_bit_exact(outs[k].download(ra.RESULT_DTYPE, n), refs[k % 2], "back to back, round %d, batch %d" % (rnd, k))
        for k in range(4):                       # ... and alone again
            ctx.intersect_device(scene, env, d_sets[k % 2].ptr, outs[0].ptr, n, lane=ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
            assert_bit_exact(outs[0].download(ra.RESULT_DTYPE, n), refs[k % 2], "lone batch again %d" % k)
        for d in d_sets + outs:
            d.free()
        scene.destroy(); env.destroy()
