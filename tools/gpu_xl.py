"""battlefield-synth at sizes past the 256 MiB Infinity Cache: kernel time, oracle parity on a sample, algorithmic bytes, and (under
rocprofv3 --pmc) the fabric traffic of one isolated launch.   python tools/gpu_xl.py [grid] [rays: diffuse|random] [variant] [iters]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 3400
kind = sys.argv[2] if len(sys.argv) > 2 else "diffuse"
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 12
t0 = time.time()
sc = synth.battlefield_synth(grid=grid, boxes=int(4096 * (grid / 700.0) ** 2) // 4 * 4, quads=int(20000 * (grid / 700.0) ** 2))
t1 = time.time()
host = ra.HostScene(sc["vertices"], sc["indices"])
t2 = time.time()
with ra.Context(device=0, kernel_variant=variant, lanes=1, chain_launches=2) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])
    cam = sc["camera"]
    if kind == "aerial":       # a view from above that covers the whole terrain: the first bounce then starts everywhere in the scene
        cam = dict(origin=np.array([0.0, 420.0, 0.0], np.float32), target=np.array([0.0, 0.0, 0.0], np.float32), up=np.array([0.0, 0.0, 1.0], np.float32), fov=26.0)
    prim, _ = synth.primary_rays(cam, 1024, 1024)
    hits = ctx.intersect(scene, env, prim)
    rays = synth.random_rays(1 << 20, 7) if kind == "random" else synth.diffuse_bounce_rays(sc, prim, hits, 1 << 20)
    n = len(rays)
    d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(rays)
    ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, iters)
    got = d_o.download(ra.RESULT_DTYPE, n)
    t3 = time.time()
    ref, nv, npairs, _ = orc.traverse(host.blobs(), rays, env=sc["env"], counters=True)
    t4 = time.time()
    alg = orc.algorithmic_bytes(ref, nv, npairs)
    hit = ref["triangle"] != 0xFFFFFFFF
    same = bool(np.array_equal(got["triangle"], ref["triangle"]) and all(np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)) for f in ("t", "u", "v")))
    msm = float(np.mean(ms[len(ms) // 2:]))
    print(json.dumps(dict(grid=grid, kind=kind, variant=variant, triangles=len(sc["indices"]), node_mb=round(host.nodes.nbytes / 1e6, 1), pair_mb=round(host.pairs.nbytes / 1e6, 1),
                          gen_s=round(t1 - t0, 1), build_s=round(t2 - t1, 1), oracle_s=round(t4 - t3, 1), height=scene.info["inner_height"], spill=scene.info["spill_levels"],
                          nv=round(float(nv.mean()), 2), np=round(float(npairs.mean()), 2), hit=round(float(hit.mean()), 3), alg_bytes=int(alg),
                          ms_first=round(ms[0], 4), ms=round(msm, 4), mrays=round(n / msm / 1e3, 1), alg_gbs=round(alg / msm / 1e6, 1), alg_frac_hbm=round(alg / msm / 1e6 / 8000, 4),
                          bit_exact=same)), flush=True)
    scene.destroy(); env.destroy(); d_r.free(); d_o.free()
