"""Drain-splitting kernel (V5) diagnostics: donations / fallbacks / iterations per launch, timing against V2."""
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
diff = synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20)
opts = [json.loads(a) for a in sys.argv[1:]] or [dict(kernel_variant=26), dict(kernel_variant=32)]
for opt in opts:
    with ra.Context(device=0, **opt) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        for n in (1 << 16, 1 << 20):
            d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(diff[:n])
            ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 2); ctx.read_stats()
            ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 5)
            st = ctx.read_stats()
            raw = st.get("raw", None)
            print(json.dumps(dict(opt=opt, n=n, ms=round(float(np.median(ms)), 4), inner_iters_per_wave=round(st["inner_iters"] / max(1, st["waves"]), 1),
                                  leaf_iters_per_wave=round(st["leaf_iters"] / max(1, st["waves"]), 1), waves=st["waves"] // 5,
                                  extra={k: v for k, v in st.items() if k not in ("inner_iters", "leaf_iters", "waves")})), flush=True)
            d_r.free(); d_o.free()
        scene.destroy(); env.destroy()
