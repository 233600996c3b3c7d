"""GPU test of the callers either side of the path: the racc:: C++ interface (include/RayAccelerator.h,
rayaccel_amd/csrc/racc_api.cpp) driven the way the reference's example app drives it — scene file ->
createScene / createEnvironment -> render(spawn, shade) with 128x128 primary tiles and bounces — with every
traced ray re-traced by the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

import rayaccel_amd as ra
from oracle import oracle as orc
from rayaccel_amd import synth
from helpers import MISS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "render_check")
REC = np.dtype([("pixel", "<u4"), ("depth", "<u4"), ("ray", orc.RAY_DTYPE), ("res", orc.RESULT_DTYPE)])


def _run(tmp_path, sc, w, h, depth, frames=1, env=None):
    scene_file = os.path.join(tmp_path, "scene.bin")
    out_file = os.path.join(tmp_path, "out.bin")
    synth.write_scene_bin(scene_file, sc)
    e = dict(os.environ, **(env or {}))
    p = subprocess.run([BIN, scene_file, out_file, str(w), str(h), str(depth), str(frames)], capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode == 0, p.stdout + p.stderr
    info = json.loads(p.stdout.strip().splitlines()[-1])
    raw = open(out_file, "rb").read()
    count, traced = np.frombuffer(raw[:16], "<u8")
    recs = np.frombuffer(raw[16:], REC)
    assert len(recs) == count == traced == info["raysTraced"]
    return info, recs


@pytest.mark.parametrize("cfg", [dict(), dict(RACC_CPU_THREADS="3", RACC_BATCH="20000"),
                                 # starved scheduler: tiny streams, few rays in flight, one lane, small shade batches
                                 dict(RACC_CPU_THREADS="5", RACC_BATCH="1024", RACC_IN_FLIGHT="40000", RACC_GPU_THREADS="1", RACC_SHADE_BATCH="300"),
                                 dict(RACC_CPU_THREADS="1", RACC_BATCH="16384", RACC_IN_FLIGHT="16384", RACC_GPU_THREADS="4"),
                                 dict(RACC_BUILD_QUALITY="0")])      # the reference builder's tree (racc::createScene builds quality 1 by default)
def test_render_matches_oracle(tmp_path, small_scene, small_host, small_default_host, cfg):
    assert ra.RAY_DTYPE.itemsize == 32
    info, recs = _run(str(tmp_path), small_scene, 512, 384, 3, frames=2, env=cfg)
    prim = recs[recs["depth"] == 0]
    assert len(prim) == (512 // 128) * (384 // 128) * 128 * 128            # every tile spawned exactly once per frame
    assert np.array_equal(np.sort(prim["pixel"]), np.sort(synth.primary_rays(small_scene["camera"], 512, 384)[1]))
    hits_d0 = int((prim["res"]["triangle"] != MISS).sum())
    assert int((recs["depth"] == 1).sum()) == hits_d0                      # one bounce per depth-0 hit, none lost
    assert recs["depth"].max() == 2
    blobs = (small_host if cfg.get("RACC_BUILD_QUALITY") == "0" else small_default_host).blobs()      # what racc::createScene built in that run
    ref = orc.traverse(blobs, np.ascontiguousarray(recs["ray"]), env=small_scene["env"], threads=8)
    got = np.ascontiguousarray(recs["res"])
    assert np.array_equal(got["triangle"], ref["triangle"])
    hit = ref["triangle"] != MISS
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32))
        np.testing.assert_allclose(got[f][~hit], ref[f][~hit], rtol=1e-5, atol=1e-5)


def test_fast_traversal_mode_through_render(tmp_path, small_scene, small_default_host):
    """racc::setFastTraversal / RACC_FAST_TRAVERSAL=1: racc::render on the compressed 4-wide kernel (kernel_variant 50).  Every traced ray
    re-traced by the oracle: the same closest hit, bit for bit, except exact-distance ties and arbiter-confirmed closer hits
    (tests/helpers.py::assert_same_closest_hit) — the documented contract of the opt-in mode; the default stays bit-exact."""
    from helpers import assert_same_closest_hit
    info, recs = _run(str(tmp_path), small_scene, 512, 384, 3, frames=1, env=dict(RACC_FAST_TRAVERSAL="1"))
    rays = np.ascontiguousarray(recs["ray"])
    ref = orc.traverse(small_default_host.blobs(), rays, env=small_scene["env"], threads=8)
    differing = assert_same_closest_hit(np.ascontiguousarray(recs["res"]), ref, "fast traversal through racc::render",
                                        arbiter=dict(vertices=small_scene["vertices"], indices=small_scene["indices"], rays=rays))
    print("fast traversal: %d of %d records differ from the oracle (ties / arbiter-confirmed closer hits)" % (differing, len(rays)))
    assert len(recs) > 3 * 128 * 128


def _check_against_oracle(recs, blobs, env):
    ref = orc.traverse(blobs, np.ascontiguousarray(recs["ray"]), env=env, threads=8)
    got = np.ascontiguousarray(recs["res"])
    assert np.array_equal(got["triangle"], ref["triangle"])
    hit = ref["triangle"] != MISS
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32))
        np.testing.assert_allclose(got[f][~hit], ref[f][~hit], rtol=1e-5, atol=1e-5)
    return ref


def test_one_context_over_two_engine_contexts(tmp_path, small_scene, small_default_host):
    """Multi-device behind the boundary (racc::gpuContextForDevices), rehearsed on the one GPU of the test box with the entry
    list 0,0: two engine contexts, scene and environment replicated, ray streams sharded over them as whole streams.  Same
    rays, same results, none lost or duplicated."""
    info, recs = _run(str(tmp_path), small_scene, 512, 384, 3, frames=2, env=dict(RACC_DEVICES="0,0", RACC_BATCH="8192"))
    info1, recs1 = _run(str(tmp_path), small_scene, 512, 384, 3, frames=2, env=dict(RACC_BATCH="8192"))
    assert len(recs) == len(recs1)
    _check_against_oracle(recs, small_default_host.blobs(), small_scene["env"])
    key = lambda r: np.lexsort((r["ray"]["dir"][:, 2], r["ray"]["dir"][:, 0], r["depth"], r["pixel"]))
    a, b = recs[key(recs)], recs1[key(recs1)]
    assert a.tobytes() == b.tobytes()              # the same set of (pixel, depth, ray, result) records as with one engine context


def test_config0_64k_primary_rays_on_the_full_scene(tmp_path, full, full_default_blobs):
    """BASELINE configs[0]: battlefield-synth (1.07 M triangles), 256x256 pinhole-coherent primary rays through racc::render
    (spawn tiles of 128x128 as TiledRenderer.cpp:55-67 does; the reference runs this config on its CPU path, which needs Embree —
    here the same plumbing feeds the GPU).  Every ray re-traced by the oracle."""
    sc = full["sc"]
    info, recs = _run(str(tmp_path), sc, 256, 256, 1, frames=1)
    assert len(recs) == 65536 and info["raysTraced"] == 65536
    assert np.array_equal(np.sort(recs["pixel"]), np.arange(65536, dtype=np.uint32))
    ref = _check_against_oracle(recs, full_default_blobs, sc["env"])
    assert (ref["triangle"] != MISS).mean() > 0.3


def test_config4_1080p_ray_streams_retraced_by_the_oracle(tmp_path, full, full_default_blobs):
    """BASELINE configs[4]'s ray streams at full size: 1920x1080 primaries (15x8 tiles = 1,966,080 rays) plus their first
    bounce through racc::render on battlefield-synth; every one of the ~3 M traced rays re-traced by the oracle."""
    sc = full["sc"]
    info, recs = _run(str(tmp_path), sc, 1920, 1080, 2, frames=1)
    prim = recs[recs["depth"] == 0]
    assert len(prim) == 15 * 8 * 128 * 128
    assert int((recs["depth"] == 1).sum()) == int((prim["res"]["triangle"] != MISS).sum()) > 500000
    _check_against_oracle(recs, full_default_blobs, sc["env"])


def test_create_context_without_gpu_context_fails_loudly(tmp_path):
    src = os.path.join(str(tmp_path), "t.cpp")
    open(src, "w").write('#include "RayAccelerator.h"\nint main(){ racc::init(); racc::Configuration c = racc::defaultConfiguration(nullptr);'
                         ' return racc::createContext(c) == nullptr ? 0 : 1; }\n')
    exe = os.path.join(str(tmp_path), "t")
    lib = os.path.join(ROOT, "rayaccel_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", lib,
                           "-lrayaccelerator", "-lracc_hip", "-Wl,-rpath," + lib])
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0 and "RayAccelerator:" in p.stderr


def test_null_callbacks_scheduler_rate(tmp_path, small_scene):
    """render_check --null-callbacks: racc::render with callbacks that cost nothing (spawn = one memcpy of a pre-generated tile, shade
    consumes and emits nothing) — the scheduler + host RayStream path alone (what bench.py reports as `scheduler_only_null_callbacks`).
    Every spawned ray is traced and handed to shade exactly once; with RACC_BUILD_QUALITY=0 racc::createScene builds the reference builder's tree."""
    scene_file = os.path.join(str(tmp_path), "scene.bin")
    synth.write_scene_bin(scene_file, small_scene)
    for env in (dict(), dict(RACC_BUILD_QUALITY="0"), dict(RACC_CPU_THREADS="2", RACC_GPU_THREADS="1", RACC_BATCH="4096")):
        p = subprocess.run([BIN, scene_file, "--null-callbacks", "1024", "512", "3", "3"], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stdout + p.stderr
        info = json.loads(p.stdout.strip().splitlines()[-1])
        assert info["raysTraced"] == info["consumed"] == 8 * 4 * 16384 * 3
        assert info["mrays_per_s_best"] > 5.0


def test_fast_callbacks_cannot_starve_the_shading_of_output_streams(tmp_path, full):
    """Round 5: with callbacks that cost nothing the scheduler deadlocked about once in eight frames-sets — every ray stream had been flushed
    to the GPU partly filled, came back traced, and no stream was left to shade INTO (shade needs an output stream, reference
    RayAccelerator.cpp:108; the reference's stream count only covers streams that leave the fill lists full).  The flush now leaves the
    emptiest streams behind (racc_api.cpp, gpuWorker).  Stream batch of 262,144 rays / six submission threads at 1080p were the
    configurations that hung; several runs each, under the frame watchdog (a hang aborts with the scheduler's state instead of timing out)."""
    scene_file = os.path.join(str(tmp_path), "scene.bin")
    synth.write_scene_bin(scene_file, full["sc"], viewport=(1920, 1080))
    for cfg in (dict(RACC_BATCH="262144"), dict(RACC_GPU_THREADS="6")):
        for run in range(6):
            p = subprocess.run([BIN, scene_file, "--null-callbacks", "1920", "1080", "16", "4"], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, RACC_CPU_THREADS="16", RACC_RENDER_WATCHDOG_S="5", **cfg))
            assert p.returncode == 0, "%s run %d: %s" % (cfg, run, (p.stdout + p.stderr)[-1500:])
            info = json.loads(p.stdout.strip().splitlines()[-1])
            assert info["raysTraced"] == info["consumed"] == 15 * 8 * 16384 * 16
