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
big = np.concatenate([synth.diffuse_bounce_rays(sc, rays, ref, 1 << 20, first_sample=s) for s in range(8)])
combos = [dict(kernel_variant=9), dict(kernel_variant=12), dict(kernel_variant=12, leaf_min=16), dict(kernel_variant=12, leaf_min=4), dict(kernel_variant=12, refill_min=16), dict(kernel_variant=12, refill_min=48)]
if len(sys.argv) > 1:
    combos = [json.loads(a) for a in sys.argv[1:]]
for opt in combos:
    with ra.Context(device=0, **opt) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        d_r = ctx.alloc(big.nbytes); d_o = ctx.alloc(len(big) * 16); d_r.upload(big)
        for n in (1 << 20, 1 << 23):
            ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 1); ctx.read_stats()
            ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 3)
            st = ctx.read_stats()
            it = st["inner_iters"] + st["leaf_iters"] + st["refill_iters"]
            print(json.dumps(dict(opt=opt, n=n, ms=round(float(np.median(ms)), 4),
                                  inner_util=round(st["inner_lanes"] / max(1, st["inner_iters"]) / 64, 3), leaf_util=round(st["leaf_lanes"] / max(1, st["leaf_iters"]) / 64, 3),
                                  iters_per_ray_x64=round(it / 3 / n * 64, 1), inner=round(st["inner_iters"] / 3 / n * 64, 1), leaf=round(st["leaf_iters"] / 3 / n * 64, 1), refill=round(st["refill_iters"] / 3 / n * 64, 2),
                                  frac_inner=round(st["cy_inner"] / st["cy_wave"], 3), frac_leaf=round(st["cy_leaf"] / st["cy_wave"], 3), frac_refill=round(st["cy_refill"] / st["cy_wave"], 3),
                                  cyc_inner=round(st["cy_inner"] / max(1, st["inner_iters"])), cyc_leaf=round(st["cy_leaf"] / max(1, st["leaf_iters"])), cyc_refill=round(st["cy_refill"] / max(1, st["refill_iters"])),
                                  frac_shuffle=round(st.get("cy_shuffle", 0) / st["cy_wave"], 3), cyc_shuffle=round(st.get("cy_shuffle", 0) / max(1, st.get("shuffles", 1))), shuffles_per_wave=round(st.get("shuffles", 0) / 3 / max(1, st["waves"] / 3), 1))), flush=True)
        scene.destroy(); env.destroy(); d_r.free(); d_o.free()
