"""Overlapped launches (lane = AUTO) vs one at a time, per kernel variant / options:  python tools/gpu_overlap.py 43 44 '{"kernel_variant":44,"lanes":2}'"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
diff = synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20)
want = None
for arg in sys.argv[1:] or ["43"]:
    v = json.loads(arg)
    opt = dict(lanes=4)
    opt.update(v if isinstance(v, dict) else dict(kernel_variant=v))
    with ra.Context(device=0, **opt) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        n = len(diff)
        d_r = ctx.alloc(n * 32); d_r.upload(diff)
        outs = [ctx.alloc(n * 16) for _ in range(ctx.lanes)]
        row = dict(opt=opt)
        for name, lane_of in (("serial", lambda k: 0), ("overlapped", lambda k: ra.LANE_AUTO)):
            for k in range(6):
                ctx.intersect_device(scene, env, d_r.ptr, outs[k % ctx.lanes].ptr, n, lane=lane_of(k))
            ctx.wait(ra.LANE_AUTO)
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for k in range(40):
                    ctx.intersect_device(scene, env, d_r.ptr, outs[k % ctx.lanes].ptr, n, lane=lane_of(k))
                ctx.wait(ra.LANE_AUTO)
                best = min(best, (time.perf_counter() - t0) / 40)
            row[name + "_ms"] = round(best * 1e3, 4)
            row[name + "_mrays"] = round(n / best / 1e6, 1)
        got = outs[0].download(orc.RESULT_DTYPE, n).tobytes()
        if want is None:
            want = got
        assert got == want, "results differ"
        print(json.dumps(row), flush=True)
        scene.destroy(); env.destroy(); d_r.free(); [o.free() for o in outs]
