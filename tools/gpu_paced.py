"""Paced issue: 1M-ray device batches issued every P microseconds instead of back to back.  With the lazy chain a batch issued
while the chain's kernels are in their drain is only PUBLISHED to them — traced by the few long-ray waves still alive.  Measured in round 6, with
and without a finer liveness rule (drain tracking: built, no effect, removed again; both runs in profiles/r06/paced_issue.txt): within 1.17 x the floor.
   python tools/gpu_paced.py            -> one JSON line per (chain mode, pause): ms for 24 batches, the GPU-bound floor beside it"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=None)
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
hits = orc.traverse(host.blobs(), prim, threads=16)
rays = synth.diffuse_bounce_rays(sc, prim, hits, 1 << 20)
ref = orc.traverse(host.blobs(), rays, env=sc["env"], threads=16)
N = 24
for mode, opt in (("lazy chain", dict()), ("chain, a kernel per launch", dict(chain_launches=3)), ("no chain", dict(chain_launches=2))):
    with ra.Context(device=0, **opt) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        d_r = ctx.alloc(rays.nbytes); d_r.upload(rays)
        outs = [ctx.alloc(len(rays) * 16) for _ in range(N)]
        def sequence(pause):
            ctx.synchronize()
            t0 = time.perf_counter()
            nxt = t0
            for o in outs:
                ctx.intersect_device(scene, env, d_r.ptr, o.ptr, len(rays), lane=ra.LANE_AUTO)
                nxt += pause
                while pause and time.perf_counter() < nxt:
                    pass
            ctx.wait(ra.LANE_AUTO)
            return time.perf_counter() - t0
        sequence(0.0)
        b2b = min(sequence(0.0) for _ in range(5))
        for pause_us in (0, 100, 150, 200, 250, 300, 400, 600, 1000):
            t = min(sequence(pause_us * 1e-6) for _ in range(5))
            bad = sum(int(o.download(orc.RESULT_DTYPE, len(rays))["triangle"].tobytes() != ref["triangle"].tobytes()) for o in outs[::7])
            print(json.dumps(dict(mode=mode, pause_us=pause_us, ms=round(t * 1e3, 3), floor_ms=round(max(b2b, N * pause_us * 1e-6) * 1e3, 3),
                                  over_floor=round(t / max(b2b, N * pause_us * 1e-6), 3), wrong_batches=bad)), flush=True)
        for o in outs: o.free()
        d_r.free(); scene.destroy(); env.destroy()
