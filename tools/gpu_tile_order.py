"""What would loading coherent primaries in 8x8 pixel blocks per wave (instead of the batch's 64x1 strips) be worth?  The 1M primary batch
(128-wide tiles, row-major inside a tile, TiledRenderer.cpp:55-67 + Camera.cpp:60-67) as is and re-ordered on the host so that every 64
consecutive rays are an 8x8 block; same rays, same kernel.   python tools/gpu_tile_order.py [variant ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
n, W = len(prim), 128
idx = np.arange(n); band = idx // (8 * W); j = idx % (8 * W); block = j // 64; l = j % 64
src = band * 8 * W + (l // 8) * W + (block % (W // 8)) * 8 + (l % 8)
blocks = np.ascontiguousarray(prim[src])
for v in [int(a) for a in sys.argv[1:]] or [0, 50]:
    with ra.Context(device=0, kernel_variant=v) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap); env = ctx.create_environment(sc["env"])
        row = dict(variant=v)
        for name, rays in (("strips_64x1", prim), ("blocks_8x8", blocks)):
            d_r = ctx.alloc(n * 32); d_r.upload(rays); outs = [ctx.alloc(n * 16) for _ in range(3)]
            ctx.intersect_device_timed(scene, env, d_r.ptr, outs[0].ptr, n, 20)
            row[name + "_ms"] = round(float(np.median(ctx.intersect_device_timed(scene, env, d_r.ptr, outs[0].ptr, n, 20))), 4)
            for k in range(6): ctx.intersect_device(scene, env, d_r.ptr, outs[k % 3].ptr, n, lane=ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
            t0 = time.perf_counter()
            for k in range(60): ctx.intersect_device(scene, env, d_r.ptr, outs[k % 3].ptr, n, lane=ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
            row[name + "_back_to_back_mrays"] = round(60 * n / (time.perf_counter() - t0) / 1e6, 1)
            d_r.free(); [o.free() for o in outs]
        print(row, flush=True)
        scene.destroy(); env.destroy()
