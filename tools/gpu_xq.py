"""A/B of kernel variants / launch options on four batches (results must stay bit-identical).
   python tools/gpu_xq.py 22 25 '{"tail_rays": 131072}'"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
diff = np.concatenate([synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20, first_sample=s) for s in range(4)])
base = {}
variants = [json.loads(v) for v in sys.argv[1:]] or [22, 25, 22, 25]      # an int = kernel_variant, a dict = Context options
for v in variants:
    with ra.Context(device=0, **(v if isinstance(v, dict) else dict(kernel_variant=v))) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        row = dict(opt=v)
        for name, rays, n in (("primary_1M", prim, 1 << 20), ("diffuse_1M", diff, 1 << 20), ("diffuse_4M", diff, 1 << 22), ("diffuse_64K", diff, 1 << 16)):
            d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(rays[:n])
            ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 3)
            ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 20)
            row[name] = round(float(np.median(ms)), 4)
            out = d_o.download(orc.RESULT_DTYPE, n)
            key = (name,)
            if key in base:
                assert out.tobytes() == base[key], "%s differs on %s" % (v, name)
            else:
                base[key] = out.tobytes()
            d_r.free(); d_o.free()
        print(json.dumps(row), flush=True)
        scene.destroy(); env.destroy()
