"""4-wide kernels (V9) against the BVH2-order kernel (V8) and the oracle: result differences classified, timings.
   python tools/gpu_wide.py 45 46 '{"kernel_variant":45,"inner_reps":2}'"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref_prim = orc.traverse(host.blobs(), prim, env=sc["env"], threads=16)
diff = np.concatenate([synth.diffuse_bounce_rays(sc, prim, ref_prim, 1 << 20, first_sample=s) for s in range(4)])
ref_diff = orc.traverse(host.blobs(), diff[:1 << 20], env=sc["env"], threads=16)


def classify(got, want):
    """(differing records, of which exact-distance ties, real differences)"""
    bad = np.nonzero((got["triangle"] != want["triangle"]) | (got["t"].view("<u4") != want["t"].view("<u4")) |
                     (got["u"].view("<u4") != want["u"].view("<u4")) | (got["v"].view("<u4") != want["v"].view("<u4")))[0]
    hit = (got["triangle"][bad] != 0xFFFFFFFF) & (want["triangle"][bad] != 0xFFFFFFFF)
    tie = hit & (np.abs(got["t"][bad] - want["t"][bad]) <= 1e-6 * np.abs(want["t"][bad]))
    miss_rgb = (got["triangle"][bad] == 0xFFFFFFFF) & (want["triangle"][bad] == 0xFFFFFFFF) & \
               (np.abs(got["t"][bad] - want["t"][bad]) < 1e-5) & (np.abs(got["u"][bad] - want["u"][bad]) < 1e-5) & (np.abs(got["v"][bad] - want["v"][bad]) < 1e-5)
    return len(bad), int(tie.sum()), int(len(bad) - tie.sum() - miss_rgb.sum())


for arg in sys.argv[1:] or ["43", "45", "46"]:
    v = json.loads(arg)
    opt = dict(lanes=4)
    opt.update(v if isinstance(v, dict) else dict(kernel_variant=v))
    with ra.Context(device=0, **opt) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        row = dict(opt=opt)
        for name, rays, n, want in (("primary_1M", prim, 1 << 20, ref_prim), ("diffuse_1M", diff, 1 << 20, ref_diff), ("diffuse_4M", diff, 1 << 22, None), ("diffuse_64K", diff, 1 << 16, ref_diff[:1 << 16])):
            d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(rays[:n])
            ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 3)
            ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 20)
            row[name] = round(float(np.median(ms)), 4)
            if want is not None:
                row[name + "_diff_tie_real"] = classify(d_o.download(orc.RESULT_DTYPE, n), want)
            d_r.free(); d_o.free()
        n = 1 << 20
        d_r = ctx.alloc(n * 32); d_r.upload(diff[:n])
        outs = [ctx.alloc(n * 16) for _ in range(ctx.lanes)]
        for k in range(6):
            ctx.intersect_device(scene, env, d_r.ptr, outs[k % ctx.lanes].ptr, n, lane=ra.LANE_AUTO)
        ctx.wait(ra.LANE_AUTO)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for k in range(40):
                ctx.intersect_device(scene, env, d_r.ptr, outs[k % ctx.lanes].ptr, n, lane=ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
            best = min(best, (time.perf_counter() - t0) / 40)
        row["overlapped_ms"] = round(best * 1e3, 4)
        row["overlapped_mrays"] = round(n / best / 1e6, 1)
        print(json.dumps(row), flush=True)
        scene.destroy(); env.destroy(); d_r.free(); [o.free() for o in outs]
