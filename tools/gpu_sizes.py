"""Single-launch time per batch size, per kernel variant (where the 4-wide kernel's shorter dependent chain pays):
   python tools/gpu_sizes.py 43 45"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=int(os.environ.get("RACC_SWEEP_QUALITY", "1")))
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
diff = np.concatenate([synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20, first_sample=s) for s in range(2)])
for arg in sys.argv[1:] or ["43", "45"]:
    v = json.loads(arg)
    opt = v if isinstance(v, dict) else dict(kernel_variant=v)
    with ra.Context(device=0, **opt) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        row = dict(opt=opt)
        for lg in range(12, 22):
            n = 1 << lg
            d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(diff[:n])
            ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 3)
            row["2^%d" % lg] = round(float(np.median(ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 15))), 4)
            d_r.free(); d_o.free()
        print(json.dumps(row), flush=True)
        scene.destroy(); env.destroy()
