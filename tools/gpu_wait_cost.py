import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc
sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
rays = synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20)
for opts in ({}, {"chain_launches": 2}):
    with ra.Context(device=0, **opts) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap); env = ctx.create_environment(sc["env"])
        d_r = ctx.alloc(rays.nbytes); d_r.upload(rays); d_o = [ctx.alloc((1 << 20) * 16) for _ in range(4)]
        for K in (1, 4):
            issue, wait = [], []
            for rep in range(12):
                ctx.synchronize()
                t0 = time.perf_counter()
                for k in range(K): ctx.intersect_device(scene, env, d_r.ptr, d_o[k].ptr, 1 << 20, lane=ra.LANE_AUTO)
                t1 = time.perf_counter()
                time.sleep(0.005)      # the GPU has long finished
                t2 = time.perf_counter()
                ctx.wait(ra.LANE_AUTO)
                t3 = time.perf_counter()
                issue.append((t1 - t0) * 1e6); wait.append((t3 - t2) * 1e6)
            print(json.dumps(dict(opts=opts, K=K, issue_us_median=round(float(np.median(issue[2:])), 1), wait_on_idle_gpu_us_median=round(float(np.median(wait[2:])), 1), wait_min=round(min(wait[2:]), 1))), flush=True)
