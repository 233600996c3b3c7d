"""What a sequence of K chained 1M-ray batches costs beyond K times the steady-state batch: ms per sequence for K = 1 .. 64, per engine
   options (best and median of 7), and the line fitted through K >= 4 — its intercept is the fixed cost of a sequence (ramp-up, drain, the
   host's first launch and last wait), its slope the steady-state batch.
   python tools/gpu_chain_k.py '{}' '{"waves_per_simd": 4}' ...

   --vs-single: What does chaining cost against ONE launch over the same rays?  64M first-bounce diffuse rays (8 sample sets, tiled) traced as one launch,
as 64 chained 1M-ray batches and as 16 chained 4M-ray batches, with and without a probe image (the lazy chain's kernels sample it in
their epilogue).   python tools/gpu_chain_k.py --vs-single"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc
import ctypes as C


def chain_k(argv):
    sc = synth.battlefield_synth()
    host = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    ref = orc.traverse(host.blobs(), prim, threads=16)
    sets = [synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20, first_sample=s) for s in range(8)]
    variants = [json.loads(v) for v in argv] or [{}]
    for v in variants:
        with ra.Context(device=0, **v) as ctx:
            scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
            env = ctx.create_environment(sc["env"])
            d_sets = []
            for s in sets:
                d = ctx.alloc(s.nbytes); d.upload(s); d_sets.append(d)
            d_outs = [ctx.alloc((1 << 20) * 16) for _ in range(8)]
            row = dict(opt=v, ms={})
            ks = (1, 2, 4, 8, 12, 16, 20, 32, 64)
            for steps in ks:
                ts = []
                for rep in range(8):
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    for k in range(steps):
                        ctx.intersect_device(scene, env, d_sets[k % 8].ptr, d_outs[k % 8].ptr, 1 << 20, lane=ra.LANE_AUTO)
                    ctx.wait(ra.LANE_AUTO)
                    ctx.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                ts = sorted(ts[1:])
                row["ms"][steps] = [round(ts[0], 4), round(ts[len(ts) // 2], 4)]
            x = np.array([k for k in ks if k >= 4], float); y = np.array([row["ms"][int(k)][0] for k in x])
            slope, icpt = np.polyfit(x, y, 1)
            row["fit_best"] = dict(ms_per_batch=round(float(slope), 4), fixed_ms=round(float(icpt), 4), mrays_per_s_k20=round(20 * (1 << 20) / row["ms"][20][0] / 1e3, 1))
            print(json.dumps(row), flush=True)
            for d in d_sets + d_outs: d.free()
            scene.destroy(); env.destroy()


def vs_single():
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


if __name__ == "__main__":
    if "--vs-single" in sys.argv: vs_single()
    else: chain_k(sys.argv[1:])
