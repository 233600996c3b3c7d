"""A/B of the device node order (RACC_NODE_ORDER=0: largest boxes first, then the blob's order; 1: every 128 B line = a node and the
child a ray most likely enters next) on the bench scene and on battlefield-synth-XL: single launches, chained steps, identical results.
   python tools/gpu_order_ab.py [xl_grid]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

xl_grid = int(sys.argv[1]) if len(sys.argv) > 1 else 3400
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]


def batches_for(sc, host, ctx_factory):
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    ref = orc.traverse(host.blobs(), prim, threads=16)
    diff = np.concatenate([synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20, first_sample=s) for s in range(4)])
    return dict(primary_1M=prim, diffuse_1M=diff[:1 << 20], diffuse_64k=diff[:1 << 16], diffuse_4M=diff, random_1M=synth.random_rays(1 << 20, 7))


def run(sc, host, names, label):
    b = batches_for(sc, host, None)
    base = {}
    for variant in variants:
      for order in (0, 1, 0, 1):
        os.environ["RACC_NODE_ORDER"] = str(order)
        with ra.Context(device=0, kernel_variant=variant) as ctx:
            scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
            env = ctx.create_environment(sc["env"])
            row = dict(scene=label, variant=variant, order=order)
            for name in names:
                rays = b[name]; n = len(rays)
                d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(rays)
                ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 20 if n <= (1 << 20) else 4)
                ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 30 if n <= (1 << 20) else 8)
                row[name] = round(float(np.median(ms)), 4)
                out = d_o.download(ra.RESULT_DTYPE, n).tobytes()
                if variant == variants[0]:
                    assert base.setdefault(name, out) == out, "order %d changed the results of %s" % (order, name)
                if name in ("diffuse_1M", "random_1M"):      # chained steps, as bench.py issues them
                    outs = [ctx.alloc(n * 16) for _ in range(24)]
                    for steps in (20, 200):
                        for k in range(10): ctx.intersect_device(scene, env, d_r.ptr, outs[k % 24].ptr, n, lane=ra.LANE_AUTO)
                        ctx.wait(ra.LANE_AUTO)
                        t = time.perf_counter()
                        for k in range(steps): ctx.intersect_device(scene, env, d_r.ptr, outs[k % 24].ptr, n, lane=ra.LANE_AUTO)
                        ctx.wait(ra.LANE_AUTO)
                        row["%s_chained%d_mrays" % (name, steps)] = round(steps * n / (time.perf_counter() - t) / 1e6, 1)
                    for o in outs: o.free()
                d_r.free(); d_o.free()
            print(json.dumps(row), flush=True)
            scene.destroy(); env.destroy()


sc = synth.battlefield_synth()
run(sc, ra.HostScene(sc["vertices"], sc["indices"]), ("diffuse_64k", "diffuse_1M", "primary_1M", "diffuse_4M"), "bench")
if xl_grid:
    sc = synth.battlefield_synth(grid=xl_grid, boxes=int(4096 * (xl_grid / 700.0) ** 2) // 4 * 4, quads=int(20000 * (xl_grid / 700.0) ** 2))
    run(sc, ra.HostScene(sc["vertices"], sc["indices"]), ("diffuse_1M", "random_1M"), "xl%d" % xl_grid)
