"""Scheduling-policy sweep on battlefield-synth-XL with its 1M incoherent rays (where the fabric binds): one launch alone and 40 chained.
   python tools/gpu_xl_sweep.py '{"refill_min":20}' ..."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
sc = synth.battlefield_synth_xl()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=int(os.environ.get("RACC_SWEEP_QUALITY", "1")))
rays = synth.random_rays(1 << 20, 7)
base = None
for arg in sys.argv[1:] or ["{}"]:
    opt = json.loads(arg)
    with ra.Context(device=0, **opt) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        n = len(rays)
        d_r = ctx.alloc(n * 32); d_r.upload(rays)
        outs = [ctx.alloc(n * 16) for _ in range(4)]
        ctx.intersect_device_timed(scene, None, d_r.ptr, outs[0].ptr, n, 4)
        ms = float(np.median(ctx.intersect_device_timed(scene, None, d_r.ptr, outs[0].ptr, n, 10)))
        got = outs[0].download(ra.RESULT_DTYPE, n).tobytes()
        if base is None: base = got
        for k in range(8): ctx.intersect_device(scene, None, d_r.ptr, outs[k % 4].ptr, n, lane=ra.LANE_AUTO)
        ctx.wait(ra.LANE_AUTO)
        t = time.perf_counter()
        for k in range(40):
            ctx.intersect_device(scene, None, d_r.ptr, outs[k % 4].ptr, n, lane=ra.LANE_AUTO)
            if k % 4 == 3: ctx.wait(ra.LANE_AUTO)
        ctx.wait(ra.LANE_AUTO)
        dt = (time.perf_counter() - t) / 40
        print(json.dumps(dict(opt=opt, ms_alone=round(ms, 4), back_to_back_ms=round(dt * 1e3, 4), same_results=(got == base))), flush=True)
        scene.destroy(); d_r.free(); [o.free() for o in outs]
