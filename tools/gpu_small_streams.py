"""The reference's own dispatch unit: ray streams of <= 27,648 rays (RayAccelerator.cpp:520).  Rate of many such batches issued back to
back: device-resident (chained / not chained) and host page-locked (racc_hip_intersect_async on rotating lanes; blocking calls from 4 threads).
   python tools/gpu_small_streams.py [rays per stream] [streams]"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 27648
m = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=int(os.environ.get("RACC_SWEEP_QUALITY", "1")))
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref0 = orc.traverse(host.blobs(), prim, threads=16)
pool = np.ascontiguousarray(synth.diffuse_bounce_rays(sc, prim, ref0, 1 << 21))
k_distinct = min(m, len(pool) // n)
for opts in (dict(), dict(chain_launches=2)):
    with ra.Context(device=0, **opts) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap); env = ctx.create_environment(sc["env"])
        d_pool = ctx.alloc(pool.nbytes); d_pool.upload(pool)
        d_out = ctx.alloc(k_distinct * n * 16)
        def run():
            for k in range(m):
                j = k % k_distinct
                ctx.intersect_device(scene, env, d_pool.ptr + j * n * 32, d_out.ptr + j * n * 16, n, lane=ra.LANE_AUTO)
                if k % 200 == 199: ctx.wait(ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
        run(); t = time.perf_counter(); run(); dt = time.perf_counter() - t
        print(json.dumps(dict(mode="device-resident", opts=opts, rays=n, streams=m, mrays=round(m * n / dt / 1e6, 1), us_per_stream=round(dt / m * 1e6, 1))), flush=True)
        # host page-locked streams
        outs = np.zeros((k_distinct, n), ra.RESULT_DTYPE)
        t1 = ctx.register_host(pool); t2 = ctx.register_host(outs)
        lanes = ctx.lanes
        def run_async():
            for k in range(m):
                j = k % k_distinct
                ctx.intersect_async(scene, env, pool[j * n:(j + 1) * n], outs[j], lane=k % lanes)
            ctx.wait(ra.LANE_AUTO)
        run_async(); t = time.perf_counter(); run_async(); dt = time.perf_counter() - t
        print(json.dumps(dict(mode="host async, lanes rotated", opts=opts, mrays=round(m * n / dt / 1e6, 1), us_per_stream=round(dt / m * 1e6, 1))), flush=True)
        def worker(w):
            for k in range(w, m, lanes):
                j = k % k_distinct
                ctx.intersect(scene, env, pool[j * n:(j + 1) * n], outs[j], lane=w)
        def run_threads():
            th = [threading.Thread(target=worker, args=(w,)) for w in range(lanes)]
            [x.start() for x in th]; [x.join() for x in th]
        run_threads(); t = time.perf_counter(); run_threads(); dt = time.perf_counter() - t
        print(json.dumps(dict(mode="host blocking, %d threads" % lanes, opts=opts, mrays=round(m * n / dt / 1e6, 1), us_per_stream=round(dt / m * 1e6, 1))), flush=True)
        ctx.unregister_host(t1); ctx.unregister_host(t2)
        scene.destroy(); env.destroy(); d_pool.free(); d_out.free()
