"""Pins the oracle (and the product) to the REFERENCE ITSELF: oracle/_ref/traversal_gfx950.co is the reference's own OpenCL
`traversal` kernel (RayAccelerator/Kernels.h:139-242), compiled in the build container from the source where it lies with
the reference's own build options, and run here on the MI355X through the ROCm OpenCL runtime exactly as the reference
launches it (work-groups of 8, one work-item per ray, RayAccelerator.cpp:378-404).

Bar (north_star): primId identical, t/u/v within 1e-4 relative — the reference kernel is compiled with
-cl-fast-relaxed-math and native_recip, so it is NOT bit-comparable; ties (two triangles at the same t) may resolve either
way.  Miss colours are not compared: CDNA has no image/sampler hardware (clCreateImage fails with CL_INVALID_OPERATION,
and the compiled kernel contains no image instruction), so the reference kernel cannot evaluate its probe image here."""
import numpy as np
import pytest

import rayaccel_amd as ra
from oracle import oracle as orc, ref_kernel
from rayaccel_amd import synth
from helpers import MISS, comb_scene, compare_with_reference_kernel, make_rays

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_kernel.built(), reason="oracle/_ref not built (needs /root/reference at build time)")]


_compare = compare_with_reference_kernel      # (tests/helpers.py; tests/test_gpu_quality.py and tests/test_gpu_xl.py import it from here)


def test_oracle_and_product_match_the_reference_kernel(gpu_ctx, small_scene, small_host):
    blobs = small_host.blobs()
    prim, _ = synth.primary_rays(small_scene["camera"], 256, 256)
    rays = np.concatenate([prim, synth.diffuse_bounce_rays(small_scene, prim, orc.traverse(blobs, prim), 40000), synth.random_rays(20000, seed=5, ymax=30.0)])
    reference = ref_kernel.run(blobs, rays, small_scene["env"])
    _compare(reference, orc.traverse(blobs, rays, env=small_scene["env"]), "oracle vs reference kernel")
    scene = gpu_ctx.upload_scene(small_host.nodes, small_host.pairs, small_host.remap)
    _compare(reference, gpu_ctx.intersect(scene, None, rays), "HIP engine vs reference kernel")
    scene.destroy()


def test_reference_kernel_on_hand_made_cases():
    blobs = comb_scene(40)                         # 40-deep stack: inside the reference's stack[64]
    o = np.stack([np.linspace(-20, 20, 256), np.linspace(-15, 15, 256), np.full(256, -10.0)], 1)
    rays = make_rays(o, [[0, 0, 1]] * 256)
    _compare(ref_kernel.run(blobs, rays, np.zeros((2, 2, 4), np.float32)), orc.traverse(blobs, rays), "comb")
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [5, 5, 9], [6, 5, 9], [5, 6, 9]], np.float32)
    sc = dict(vertices=np.concatenate([v, np.ones((7, 1), np.float32)], 1), indices=np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6]], np.uint32))
    quad = orc.build_scene(sc["vertices"], sc["indices"])
    kat = make_rays([[0.75, 0.25, -2], [0.25, 0.75, -2], [0.75, 0.25, 3], [5.25, 5.25, 0], [5.9, 5.9, 0], [0.3, 0.6, -2]],
                    [[0, 0, 1], [0, 0, 1], [0, 0, -1], [0, 0, 1], [0, 0, 1], [0, 0.0, 1]])
    _compare(ref_kernel.run(quad, kat, np.zeros((2, 2, 4), np.float32)), orc.traverse(quad, kat), "quad KATs")


def test_full_size_batch_against_the_reference_kernel(gpu_ctx):
    """BASELINE configs[2] at full size: the reference's kernel, the oracle and the HIP engine on the same 1M diffuse rays."""
    sc = synth.battlefield_synth()
    host = ra.HostScene(sc["vertices"], sc["indices"])
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    bounce = synth.diffuse_bounce_rays(sc, prim, orc.traverse(host.blobs(), prim, threads=8), 1 << 20)
    reference = ref_kernel.run(host.blobs(), bounce, sc["env"])
    ties = _compare(reference, orc.traverse(host.blobs(), bounce, threads=8), "oracle vs reference kernel, 1M diffuse")
    scene = gpu_ctx.upload_scene(host.nodes, host.pairs, host.remap)
    _compare(reference, gpu_ctx.intersect(scene, None, bounce), "HIP engine vs reference kernel, 1M diffuse")
    scene.destroy()
    assert ties <= 4           # observed on the MI355X: 0 (printed above)
