"""Does the traversal rate depend on whether the tree fits the L2?  Same kind of scene at three sizes, 4M diffuse rays each:
   kernel time per ray and per node visit (visits from the oracle's counters).   python tools/gpu_scene_size.py [variant]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for grid, boxes, quads in ((150, 256, 1000), (300, 1024, 4000), (700, 4096, 20000)):
    sc = synth.battlefield_synth(grid=grid, boxes=boxes, quads=quads)
    host = ra.HostScene(sc["vertices"], sc["indices"])
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    ref = orc.traverse(host.blobs(), prim, threads=16)
    diff = np.concatenate([synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20, first_sample=s) for s in range(4)])
    _, nv, npairs, _ = orc.traverse(host.blobs(), diff[:1 << 18], counters=True, threads=16)
    with ra.Context(device=0, kernel_variant=variant) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        n = len(diff)
        d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(diff)
        ctx.intersect_device_timed(scene, None, d_r.ptr, d_o.ptr, n, 3)
        ms4 = float(np.median(ctx.intersect_device_timed(scene, None, d_r.ptr, d_o.ptr, n, 10)))
        ms1 = float(np.median(ctx.intersect_device_timed(scene, None, d_r.ptr, d_o.ptr, n // 4, 10)))
        slope = (ms4 - ms1) / 3.0      # ms per 1M rays in steady state
        print(json.dumps(dict(variant=variant, triangles=len(sc["indices"]) // 3, node_mb=round(host.nodes.nbytes / 1e6, 1), pair_mb=round(host.pairs.nbytes / 1e6, 1),
                              nv=round(float(nv.mean()), 1), np=round(float(npairs.mean()), 2), ms_4M=round(ms4, 4), ms_1M=round(ms1, 4),
                              steady_grays=round(1.048576 / slope, 2), ps_per_visit=round(slope * 1e9 / 1048576 / float(nv.mean() + npairs.mean()), 1))), flush=True)
        scene.destroy(); d_r.free(); d_o.free()
