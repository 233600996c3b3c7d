"""Second-opinion parity against a SYSTEM Embree (the reference's CPU path, Scene.cpp:374-484), when the box has one
(oracle/embree_adapter.py).  Neither box of this build has: the module then skips.  north_star's criterion verbatim:
primId exact (ties at equal distance excepted, judged by the arbiter rule of SURVEY §8c), t/u/v within 1e-4 relative."""
import numpy as np
import pytest

from oracle import embree_adapter
from oracle import oracle as orc
from rayaccel_amd import synth
from helpers import MISS

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not embree_adapter.available(), reason="no system Embree on this box")]


def test_engine_matches_system_embree(gpu_ctx, small_scene, small_host, small):
    prim = small["primary"]
    rays = np.concatenate([prim, synth.diffuse_bounce_rays(small_scene, prim, orc.traverse(small["blobs"], prim), 40000)])
    emb = embree_adapter.trace(small_scene, rays, threads=4)
    got = gpu_ctx.intersect(small["scene"], None, rays)
    hit_e, hit_g = emb["triangle"] != MISS, got["triangle"] != MISS
    assert (hit_e != hit_g).sum() == 0
    both = hit_e & hit_g
    diff = both & (emb["triangle"] != got["triangle"])
    assert np.allclose(emb["t"][diff], got["t"][diff], rtol=1e-5)             # a different primId only as a tie
    same = both & ~diff
    for f in ("t", "u", "v"):
        np.testing.assert_allclose(got[f][same], emb[f][same], rtol=1e-4, atol=2e-6)
