"""Batch-size scaling probe: T(N) = a + b*N separates the fixed launch/tail cost from the steady-state rate."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
rays, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), rays, threads=16)
batches = [synth.diffuse_bounce_rays(sc, rays, ref, 1 << 20, first_sample=s) for s in range(8)]
big = np.concatenate(batches)
import json as _j
opts = [_j.loads(a) for a in sys.argv[1:]] or [dict(kernel_variant=1), dict(kernel_variant=10), dict(kernel_variant=10, tail_active=100), dict(kernel_variant=10, tail_active=32), dict(kernel_variant=10, tail_active=8), dict(kernel_variant=11)]
for opt in opts:
    with ra.Context(device=0, **opt) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        d_r = ctx.alloc(big.nbytes); d_o = ctx.alloc(len(big) * 16); d_r.upload(big)
        row = {}
        for n in (1 << 16, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23):
            ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 2)
            ms = float(np.median(ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 7)))
            row[n] = round(ms, 4)
        ns = np.array(sorted(row)); ts = np.array([row[k] for k in ns])
        b, a = np.polyfit(ns[3:], ts[3:], 1)
        print(json.dumps(dict(opt=opt, ms=row, fixed_ms=round(a, 4), per_Mray_ms=round(b * 1e6, 4), steady_mrays=round(1e3 / (b * 1e6), 1))), flush=True)
        scene.destroy(); env.destroy(); d_r.free(); d_o.free()
