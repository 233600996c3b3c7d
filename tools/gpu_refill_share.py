"""What share of a wave's lifetime is the epilogue + refill block (the C++ part of traverseKernelV8 around the assembly hot loop)?  The
   statistics build (kernel_variant 42) sums shader-clock cycles inside that block and over the wave's life.   python tools/gpu_refill_share.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
diff = np.concatenate([synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20, first_sample=s) for s in range(8)])
with ra.Context(device=0, kernel_variant=42, lanes=1, chain_launches=2) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])
    for n in (1 << 20, 1 << 23):
        d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(diff[:n])
        ctx.intersect_device(scene, env, d_r.ptr, d_o.ptr, n, lane=0); ctx.wait(0)
        ctx.read_stats(0, reset=True)
        ctx.intersect_device(scene, env, d_r.ptr, d_o.ptr, n, lane=0); ctx.wait(0)
        st = ctx.read_stats(0, reset=True)
        print(json.dumps(dict(rays=n, refills_per_64_rays=round(st["refill_iters"] * 64 / n, 2), rays_per_refill=round(st["rays_loaded"] / max(1, st["refill_iters"]), 2),
                              refill_share_of_wave_lifetime=round(st["cy_refill"] / st["cy_wave"], 4), cycles_per_refill=round(st["cy_refill"] / max(1, st["refill_iters"]), 1),
                              wave_lifetime_cycles=round(st["cy_wave"] / st["waves"], 0), dequeues=st["dequeues"], waves=st["waves"])), flush=True)
        d_r.free(); d_o.free()
