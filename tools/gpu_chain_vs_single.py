"""What does chaining cost against ONE launch over the same rays?  64M first-bounce diffuse rays (8 sample sets, tiled) traced as one launch,
as 64 chained 1M-ray batches and as 16 chained 4M-ray batches, with and without a probe image (the lazy chain's kernels sample it in
their epilogue).   python tools/gpu_chain_vs_single.py"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
hits = orc.traverse(host.blobs(), prim, threads=16)
sets = synth.diffuse_bounce_batches(sc, prim, hits, 1 << 20, range(8))
M = 1 << 20
TOTAL = 64
with ra.Context(device=0) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])
    d_r = ctx.alloc(TOTAL * M * 32)
    import ctypes as C
    for k in range(TOTAL):
        ra.engine._check(ra.load_library().racc_hip_memcpy_h2d(ctx._h, d_r.ptr + k * M * 32, sets[k % 8].ctypes.data_as(C.c_void_p), M * 32))
    d_o = ctx.alloc(TOTAL * M * 16)
    for e, name in ((env, "probe image"), (None, "no probe image")):
        ctx.intersect_device_timed(scene, e, d_r.ptr, d_o.ptr, TOTAL * M, 1)
        ms = ctx.intersect_device_timed(scene, e, d_r.ptr, d_o.ptr, TOTAL * M, 3)
        print(json.dumps({"what": "one %dM-ray launch, %s" % (TOTAL, name), "ms_per_Mray": round(float(np.min(ms)) / TOTAL, 4)}), flush=True)
        for per in (1, 4, 16):
            n = per * M
            best = 1e9
            for rep in range(3):
                ctx.wait(ra.LANE_AUTO)
                t0 = time.perf_counter()
                for k in range(TOTAL // per):
                    ctx.intersect_device(scene, e, d_r.ptr + k * n * 32, d_o.ptr + k * n * 16, n, lane=ra.LANE_AUTO)
                ctx.wait(ra.LANE_AUTO)
                best = min(best, time.perf_counter() - t0)
            print(json.dumps({"what": "%d chained %dM-ray batches, %s" % (TOTAL // per, per, name), "ms_per_Mray": round(best * 1e3 / TOTAL, 4)}), flush=True)
    scene.destroy(); env.destroy()
