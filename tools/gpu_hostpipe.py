"""Host RayStream batches issued back to back: racc_hip_intersect_async on rotating lanes (copies of batch k+1 beside the kernel of
batch k beside the copy-out of batch k-1), page-locked arrays.  Prints Grays/s and the H2D / D2H rates separately.
   python tools/gpu_hostpipe.py [batches] [rays per batch]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get('RACC_HOSTPIPE_TORCH'):
    import torch
    _t = torch.zeros(1 << 20, device='cuda'); torch.cuda.synchronize()
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref0 = orc.traverse(host.blobs(), prim, threads=16)
batches = [np.ascontiguousarray(np.concatenate([synth.diffuse_bounce_rays(sc, prim, ref0, 1 << 20, first_sample=4 * k + s) for s in range((n + (1 << 20) - 1) >> 20)])[:n]) for k in range(nb)]
refs = [orc.traverse(host.blobs(), b, env=sc["env"], threads=16) for b in batches]
outs = [np.zeros(n, ra.RESULT_DTYPE) for _ in range(nb)]
lib = ra.load_library()
for lanes in ([int(os.environ['RACC_HOSTPIPE_LANES'])] if os.environ.get('RACC_HOSTPIPE_LANES') else (2, 3, 4, 5, 6, 8)):
    with ra.Context(device=0, lanes=lanes) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        for a in batches + outs:
            assert lib.racc_hip_register_host(ctx._h, a.ctypes.data, a.nbytes) == 0
        def run():
            for k in range(nb):
                rc = lib.racc_hip_intersect_async(ctx._h, scene._h, env._h, batches[k].ctypes.data_as(C.c_void_p), outs[k].ctypes.data_as(C.c_void_p), n, k % lanes)
                assert rc == 0, lib.racc_hip_last_error()
            ctx.wait(ra.LANE_AUTO)
        run()
        best = 1e9
        for _ in range(3):
            for o in outs: o[:] = 0
            t = time.perf_counter(); run(); best = min(best, time.perf_counter() - t)
        ok = all(np.array_equal(o["triangle"], r["triangle"]) and np.array_equal(o["t"][r["triangle"] != 0xFFFFFFFF].view(np.uint32), r["t"][r["triangle"] != 0xFFFFFFFF].view(np.uint32)) for o, r in zip(outs, refs))
        print(json.dumps(dict(lanes=lanes, batches=nb, rays=n, grays=round(nb * n / best / 1e9, 3), h2d_gbs=round(nb * n * 32 / best / 1e9, 1), d2h_gbs=round(nb * n * 16 / best / 1e9, 1), bit_exact=bool(ok))), flush=True)
        # the blocking, sliced entry for comparison (one batch at a time)
        if lanes == 3:
            ctx.intersect(scene, env, batches[0], outs[0])
            t = time.perf_counter()
            for k in range(nb): ctx.intersect(scene, env, batches[k], outs[k])
            dt = time.perf_counter() - t
            print(json.dumps(dict(blocking_sliced=True, grays=round(nb * n / dt / 1e9, 3))), flush=True)
        if os.environ.get("RACC_HOSTPIPE_DEBUG"):      # which step of bench.py's sequence makes its first 16-batch run take 48 ms?
            def many2(m, tag):
                t = time.perf_counter()
                stamps = []
                for k in range(m):
                    t1 = time.perf_counter()
                    lib.racc_hip_intersect_async(ctx._h, scene._h, env._h, batches[0].ctypes.data_as(C.c_void_p), outs[k % nb].ctypes.data_as(C.c_void_p), n, k % lanes)
                    stamps.append(round((time.perf_counter() - t1) * 1e3, 2))
                t1 = time.perf_counter(); ctx.wait(ra.LANE_AUTO); w = round((time.perf_counter() - t1) * 1e3, 2)
                print(json.dumps(dict(tag=tag, batches=m, total_ms=round((time.perf_counter() - t) * 1e3, 2), enqueue_ms=stamps, wait_ms=w)), flush=True)
            many2(8, "warm"); many2(16, "plain16")
            for o in outs: o[:] = 0
            many2(16, "after zeroing the result arrays")
            for _ in range(3): ctx.intersect(scene, env, batches[0], outs[0])
            many2(16, "after blocking sliced calls")
            for _ in range(3): ctx.intersect(scene, env, batches[0], outs[0])
            many2(8, "blocking, then 8"); many2(16, "then 16"); many2(64, "then 64")
        if os.environ.get("RACC_HOSTPIPE_SAME"):      # bench.py's form: ONE ray array for every batch, 8 result arrays, 8 / 16 / 64 batches in a row
            def many(m):
                t = time.perf_counter()
                for k in range(m):
                    lib.racc_hip_intersect_async(ctx._h, scene._h, env._h, batches[0].ctypes.data_as(C.c_void_p), outs[k % nb].ctypes.data_as(C.c_void_p), n, k % lanes)
                ctx.wait(ra.LANE_AUTO)
                return round((time.perf_counter() - t) * 1e3, 3)
            print(json.dumps(dict(lanes=lanes, same_ray_array_ms=[many(8), many(16), many(64), many(16), many(16), many(64)])), flush=True)
        for a in batches + outs:
            lib.racc_hip_unregister_host(ctx._h, a.ctypes.data)
        scene.destroy(); env.destroy()
