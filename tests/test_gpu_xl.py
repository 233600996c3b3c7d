"""battlefield-synth-XL (round 4): 25 M triangles = 1.3 GB of nodes, pairs and remap on the device — past the MI355X's 256 MiB
Infinity Cache, the regime in which the contractual HBM roofline can bind.  Same bar as every other config: every
record bit-identical to the oracle, and the reference's own OpenCL kernel agrees on hit/miss and primId."""
import numpy as np
import pytest

import rayaccel_amd as ra
from oracle import oracle as orc, ref_kernel
from rayaccel_amd import synth
from helpers import MISS, assert_bit_exact

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xl(gpu_ctx):
    sc = synth.battlefield_synth_xl()
    host = ra.HostScene(sc["vertices"], sc["indices"])
    scene = gpu_ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = gpu_ctx.create_environment(sc["env"])
    yield dict(sc=sc, host=host, blobs=host.blobs(), scene=scene, env=env)
    scene.destroy(); env.destroy()


def test_xl_is_past_the_infinity_cache_and_inside_the_format(xl):
    info = xl["scene"].info
    assert len(xl["sc"]["indices"]) > 24_000_000
    assert info["device_bytes"] > 4 * (256 << 20)                 # > 1 GiB resident: four Infinity Caches
    assert info["pair_count"] < (1 << 24) and info["node_count"] < (1 << 26)      # the reference format's limits (Scene.cpp:294-312)
    assert info["inner_height"] > 25 and info["spill_levels"] > 0                 # taller than the LDS part of the stack: the spill is live


def test_xl_incoherent_1M_bit_exact(gpu_ctx, xl):
    """The batch bench.py profiles as `xl`: 1M rays with origins and directions uniform over the scene — the whole 1.3 GB is the
    working set.  All 1,048,576 records against the oracle, bit for bit; lanes in rotation (chained) give the same bits."""
    rays = synth.random_rays(1 << 20, 7)
    ref = orc.traverse(xl["blobs"], rays, env=xl["sc"]["env"], threads=16)
    assert 0.5 < (ref["triangle"] != MISS).mean() < 0.9
    assert_bit_exact(gpu_ctx.intersect(xl["scene"], xl["env"], rays), ref, "XL incoherent, host path")
    d_r = gpu_ctx.alloc(rays.nbytes); d_r.upload(rays)
    outs = [gpu_ctx.alloc(len(rays) * 16) for _ in range(6)]      # one result array per batch in flight (the arrays belong to the engine until the wait)
    for o in outs:
        gpu_ctx.intersect_device(xl["scene"], xl["env"], d_r.ptr, o.ptr, len(rays), lane=ra.LANE_AUTO)
    gpu_ctx.wait(ra.LANE_AUTO)
    for o in outs:
        assert_bit_exact(o.download(ra.RESULT_DTYPE, len(rays)), ref, "XL incoherent, chained")
        o.free()
    d_r.free()


def test_xl_camera_batches_bit_exact(gpu_ctx, xl):
    sc = xl["sc"]
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    hits = gpu_ctx.intersect(xl["scene"], xl["env"], prim)
    assert_bit_exact(hits, orc.traverse(xl["blobs"], prim, env=sc["env"], threads=16), "XL primaries")
    bounce = synth.diffuse_bounce_rays(sc, prim, hits, 1 << 20)
    assert_bit_exact(gpu_ctx.intersect(xl["scene"], xl["env"], bounce), orc.traverse(xl["blobs"], bounce, env=sc["env"], threads=16), "XL diffuse")


@pytest.mark.skipif(not ref_kernel.built(), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_xl_against_the_reference_kernel(gpu_ctx, xl):
    """The reference's own OpenCL `traversal` kernel on the XL blobs (the product builder's, in the reference's format): 128k
    incoherent rays, no hit/miss disagreement, primId equal up to exact-distance ties, t/u/v within north_star's 1e-4."""
    from test_gpu_reference_kernel import _compare
    rays = synth.random_rays(1 << 17, 11)
    reference = ref_kernel.run(xl["blobs"], rays, xl["sc"]["env"])
    _compare(reference, orc.traverse(xl["blobs"], rays, threads=16), "oracle vs reference kernel, XL")
    _compare(reference, gpu_ctx.intersect(xl["scene"], None, rays), "HIP engine vs reference kernel, XL")


def test_xl_quality_tree_bit_exact(gpu_ctx, xl):
    """The same 25 M triangles through racc_host_scene_build_ex(quality = 1) — 13 M one-pair leaves, subtrees re-inserted; what bench.py's
    XL rows run on: the incoherent 1M batch, host path and chained, every record against the oracle on the same blobs; and against the
    reference builder's tree (same hits) with the visit counts that are the point of the mode."""
    sc = xl["sc"]
    host = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
    scene = gpu_ctx.upload_scene(host.nodes, host.pairs, host.remap)
    try:
        assert scene.info["pair_count"] < (1 << 24)
        rays = synth.random_rays(1 << 20, 7)
        ref, nv, npairs, _ = orc.traverse(host.blobs(), rays, env=sc["env"], counters=True, threads=16)
        got = gpu_ctx.intersect(scene, xl["env"], rays)
        assert_bit_exact(got, ref, "XL incoherent, quality 1, host path")
        d_r = gpu_ctx.alloc(rays.nbytes); d_r.upload(rays)
        outs = [gpu_ctx.alloc(len(rays) * 16) for _ in range(4)]
        for o in outs:
            gpu_ctx.intersect_device(scene, xl["env"], d_r.ptr, o.ptr, len(rays), lane=ra.LANE_AUTO)
        gpu_ctx.wait(ra.LANE_AUTO)
        for o in outs:
            assert_bit_exact(o.download(ra.RESULT_DTYPE, len(rays)), ref, "XL incoherent, quality 1, chained")
            o.free()
        d_r.free()
        base, nv0, np0, _ = orc.traverse(xl["blobs"], rays, env=sc["env"], counters=True, threads=16)
        from helpers import assert_same_hits_across_trees
        assert_same_hits_across_trees(base, ref, "XL, reference builder's tree against quality 1", t_floor=2e-5, max_other=32, uv_atol=1e-3)      # (coordinates to 490: an ulp is 3e-5)
        assert nv.mean() < 0.9 * nv0.mean() and npairs.mean() < 0.7 * np0.mean(), (nv.mean(), nv0.mean(), npairs.mean(), np0.mean())
    finally:
        scene.destroy()
