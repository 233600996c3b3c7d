"""GPU tests of the path-tracing consumer (rayaccel_amd/csrc/pathtracer.cpp ≙ Renderer/PathTracingRenderer.cpp): the
reference's own renderer is rand()-seeded and not reproducible, so the counterpart is pinned by its own
invariants: bit-reproducible frames, thread-count independence, sample-shard additivity (what the N-GPU run relies on),
radiance bounds under a constant environment, and exact primary-miss pixels."""
import os

import numpy as np
import pytest

from rayaccel_amd import synth
from rayaccel_amd.engine import path_trace, path_trace_capture_round

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene_file(tmp_path_factory, small_scene):
    p = str(tmp_path_factory.mktemp("pt") / "scene.bin")
    sc = dict(small_scene)
    env = np.zeros((16, 32, 4), np.float32)
    env[..., :3] = (0.5, 1.0, 2.0)                       # constant environment radiance
    sc["env"] = env
    synth.write_scene_bin(p, sc)
    return p


def test_reproducible_and_thread_independent(scene_file):
    a, sa = path_trace(scene_file, 256, 256, 0, 3, cpu_threads=2)
    b, sb = path_trace(scene_file, 256, 256, 0, 3, cpu_threads=7)
    assert np.array_equal(a, b) and sa["rays_traced"] == sb["rays_traced"]      # fixed-point frame: shading order is irrelevant
    assert sa["primary_rays"] == 3 * 256 * 256 and sa["rays_traced"] > sa["primary_rays"] and sa["max_depth"] == 5


def test_sample_shards_add_up(scene_file):
    whole, sw = path_trace(scene_file, 256, 256, 0, 4)
    lo, s0 = path_trace(scene_file, 256, 256, 0, 1)
    hi, s1 = path_trace(scene_file, 256, 256, 1, 3)
    assert np.array_equal(whole, lo + hi) and sw["rays_traced"] == s0["rays_traced"] + s1["rays_traced"]


def test_energy_bounds_and_primary_misses(scene_file, small_scene):
    spp = 4
    img, st = path_trace(scene_file, 256, 256, 0, spp)
    mean = img / spp
    env = np.array([0.5, 1.0, 2.0])
    # No upper bound per pixel: the reference material's expected albedo is kd + Fresnel (Materials.cpp:121-141), which
    # exceeds 1 at grazing angles, and single-sample weights reach (3F + sum kd) / 3.
    assert np.isfinite(mean).all() and (mean >= 0).all()
    sky = np.isclose(mean, env, rtol=1e-5).all(-1)
    assert 0.05 < sky.mean() < 0.95                                     # primary misses see the environment exactly
    lit = mean[~sky]
    assert 0.02 * env.mean() < lit.mean() < 1.5 * env.mean() and (lit < env).any()
    bright = mean[..., 0] > 0.01
    np.testing.assert_allclose(mean[bright][:, 1] / mean[bright][:, 0], 2.0, rtol=1e-3)          # grey materials keep the sky's colour ratio


def test_device_consumer_renders_the_same_image(scene_file):
    """pt_device.hip (generation + shading kernels, rays resident in HBM) against pathtracer.cpp (host callbacks): the same
    pt_shade.h arithmetic, the same counter RNG, an integer frame buffer -> identical frames and ray counts."""
    cpu, sc = path_trace(scene_file, 256, 256, 0, 3, shading="cpu")
    gpu, sg = path_trace(scene_file, 256, 256, 0, 3, shading="gpu")
    assert sg["rays_traced"] == sc["rays_traced"] and sg["primary_rays"] == sc["primary_rays"]
    assert np.array_equal(cpu, gpu)
    one, s1 = path_trace(scene_file, 256, 256, 0, 3, shading="gpu", samples_per_batch=1)      # batching does not matter
    assert np.array_equal(one, gpu) and s1["rays_traced"] == sg["rays_traced"]
    lo, _ = path_trace(scene_file, 256, 256, 0, 1, shading="gpu")
    hi, _ = path_trace(scene_file, 256, 256, 1, 2, shading="cpu")                                # shards from different consumers add up
    assert np.array_equal(lo + hi, gpu)


def test_config4_1080p_device_and_host_consumer_render_the_same_frame(tmp_path, full):
    """BASELINE configs[4] at full size: battlefield-synth, 1920x1080 (15x8 whole tiles, as TiledRenderer.cpp:20-22 renders),
    4 spp, depth from the scene header — the device-resident consumer against the reference-shaped host consumer: identical
    fixed-point frames and ray counts.  (64 spp is the same code 16 times over; bench.py times it.)"""
    p = os.path.join(str(tmp_path), "full.bin")
    synth.write_scene_bin(p, full["sc"], viewport=(1920, 1080))
    g, sg = path_trace(p, 1920, 1080, 0, 4, shading="gpu")
    c, sc_ = path_trace(p, 1920, 1080, 0, 4, shading="cpu")
    assert sg["primary_rays"] == sc_["primary_rays"] == 4 * 15 * 8 * 128 * 128
    assert sg["rays_traced"] == sc_["rays_traced"] > 2 * sg["primary_rays"]
    assert np.array_equal(g, c) and g[:1024].any() and not g[1024:].any()


def test_config4_1080p_64spp_equals_sixteen_shards_and_a_bounce_is_retraced(tmp_path, full, full_default_blobs):
    """BASELINE configs[4] as written: 1920x1080 x 64 spp, device-resident consumer.  The fixed-point frame and the ray count
    equal the sum of 16 shards of 4 spp (what N ranks render, tools/pathtrace.py), and one bounce of the 64-spp render — the
    rays the consumer's shading kernel generated and handed to the engine in round 5, the hit records it got back — is
    re-traced by the oracle: bit-exact (so the consumer shades what the reference's traversal would have returned)."""
    import hashlib
    from oracle import oracle as orc
    from helpers import assert_bit_exact
    p = os.path.join(str(tmp_path), "full64.bin")
    synth.write_scene_bin(p, full["sc"], viewport=(1920, 1080))
    cap = 1 << 21
    rays, hits, count = path_trace_capture_round(5, cap)
    whole, sw = path_trace(p, 1920, 1080, 0, 64, shading="gpu")
    n = int(count.value)
    assert n > 100000 and sw["primary_rays"] == 64 * 15 * 8 * 128 * 128 and sw["rays_traced"] > 2 * sw["primary_rays"]
    assert float(rays["minT"][:n].max()) > 0.0          # a secondary bounce (primaries start at minT = 0, PathTracingRenderer.cpp:410-422)
    ref = orc.traverse(full_default_blobs, rays[:n], env=full["sc"]["env"], threads=8)      # (the consumer builds its scene with the library default: the quality-1 tree)
    assert_bit_exact(hits[:n], ref, "device consumer, 64 spp, round 5 (%d rays)" % n)
    total, traced = np.zeros_like(whole), 0
    for k in range(16):
        img, st = path_trace(p, 1920, 1080, 4 * k, 4, shading="gpu")
        total += img; traced += st["rays_traced"]
    assert traced == sw["rays_traced"]
    assert hashlib.md5(total.tobytes()).hexdigest() == hashlib.md5(whole.tobytes()).hexdigest() and np.array_equal(total, whole)


def test_device_consumer_whole_tiles(scene_file):
    img, st = path_trace(scene_file, 300, 200, 0, 2, shading="gpu")
    assert (st["tiles_x"], st["tiles_y"]) == (2, 1) and st["primary_rays"] == 2 * 2 * 128 * 128 and st["threads"] == 0
    assert not img[128:].any() and not img[:, 256:].any() and img[:128, :256].any()
    empty, se = path_trace(scene_file, 100, 100, 0, 1, shading="gpu")                            # no whole tile: nothing traced
    assert not empty.any() and se["rays_traced"] == 0


def test_only_whole_tiles_are_rendered(scene_file):
    img, st = path_trace(scene_file, 300, 200, 0, 1)                    # TiledRenderer.cpp:20-22
    assert (st["tiles_x"], st["tiles_y"]) == (2, 1) and st["primary_rays"] == 2 * 128 * 128
    assert not img[128:].any() and not img[:, 256:].any() and img[:128, :256].any()


def test_sample_sharded_ranks_render_the_same_frame():
    """tools/pathtrace.py with 2 ranks (gloo rehearsal on one GPU: the ranks share device 0, frames are summed on the CPU;
    the real run is nccl, one rank per GPU): the summed frame and the ray count equal the single-process run's."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "pathtrace.py")
    common = ["--spp", "4", "--width", "512", "--height", "384", "--grid", "96"]
    one = subprocess.run([sys.executable, tool] + common, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    env = dict(os.environ, RACC_BENCH_BACKEND="gloo", RACC_BENCH_DEVICE="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", tool] + common, capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1]); b = json.loads(two.stdout.strip().splitlines()[-1])
    assert b["n_gpus"] == 2 and a["frame_md5"] == b["frame_md5"] and a["rays_traced"] == b["rays_traced"]
