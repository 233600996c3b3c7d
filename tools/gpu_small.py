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
big = synth.diffuse_bounce_rays(sc, rays, ref, 1 << 20)
r2, nv, npp, dp = orc.traverse(host.blobs(), big, counters=True)
steps = nv + npp
for n in (64, 4096, 65536, 1 << 20):
    s = steps[:n].astype(np.int64)
    wmax = s[: (n // 64) * 64].reshape(-1, 64).max(1)
    print("n=%d steps/ray mean %.1f max %d; per-64 block max: mean %.1f max %d" % (n, s.mean(), s.max(), wmax.mean(), wmax.max()))
for opt in (dict(kernel_variant=9, waves_per_simd=5),):
    with ra.Context(device=0, **opt) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        d_r = ctx.alloc(big.nbytes); d_o = ctx.alloc(len(big) * 16); d_r.upload(big)
        for n in (64, 4096, 65536, 1 << 18, 1 << 20):
            ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 2); ctx.read_stats()
            ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 5)
            st = ctx.read_stats()
            w = st["waves"] / 5
            print(json.dumps(dict(n=n, ms=round(float(np.median(ms)), 4), waves=w, inner_per_wave=round(st["inner_iters"] / 5 / w, 1), leaf_per_wave=round(st["leaf_iters"] / 5 / w, 1),
                                  refill_per_wave=round(st["refill_iters"] / 5 / w, 1), inner_util=round(st["inner_lanes"] / max(1, st["inner_iters"]) / 64, 3),
                                  cyc_per_inner=round(st["cy_inner"] / max(1, st["inner_iters"])), cyc_inner_load=round(st["cy_inner_load"] / max(1, st["inner_iters"])),
                                  cyc_per_leaf=round(st["cy_leaf"] / max(1, st["leaf_iters"])), cyc_leaf_load=round(st["cy_leaf_load"] / max(1, st["leaf_iters"])),
                                  cyc_per_refill=round(st["cy_refill"] / max(1, st["refill_iters"])), wave_life_cyc=round(st["cy_wave"] / 5 / w),
                                  frac_inner=round(st["cy_inner"] / st["cy_wave"], 3), frac_leaf=round(st["cy_leaf"] / st["cy_wave"], 3), frac_refill=round(st["cy_refill"] / st["cy_wave"], 3))), flush=True)
