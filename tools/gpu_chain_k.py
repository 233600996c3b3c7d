"""What a sequence of K chained 1M-ray batches costs beyond K times the steady-state batch: ms per sequence for K = 1 .. 64, per engine
   options (best and median of 7), and the line fitted through K >= 4 — its intercept is the fixed cost of a sequence (ramp-up, drain, the
   host's first launch and last wait), its slope the steady-state batch.
   python tools/gpu_chain_k.py '{}' '{"waves_per_simd": 4}' ..."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
sets = [synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20, first_sample=s) for s in range(8)]
variants = [json.loads(v) for v in sys.argv[1:]] or [{}]
for v in variants:
    with ra.Context(device=0, **v) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        d_sets = []
        for s in sets:
            d = ctx.alloc(s.nbytes); d.upload(s); d_sets.append(d)
        d_outs = [ctx.alloc((1 << 20) * 16) for _ in range(8)]
        row = dict(opt=v, ms={})
        ks = (1, 2, 4, 8, 12, 16, 20, 32, 64)
        for steps in ks:
            ts = []
            for rep in range(8):
                ctx.synchronize()
                t0 = time.perf_counter()
                for k in range(steps):
                    ctx.intersect_device(scene, env, d_sets[k % 8].ptr, d_outs[k % 8].ptr, 1 << 20, lane=ra.LANE_AUTO)
                ctx.wait(ra.LANE_AUTO)
                ctx.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts = sorted(ts[1:])
            row["ms"][steps] = [round(ts[0], 4), round(ts[len(ts) // 2], 4)]
        x = np.array([k for k in ks if k >= 4], float); y = np.array([row["ms"][int(k)][0] for k in x])
        slope, icpt = np.polyfit(x, y, 1)
        row["fit_best"] = dict(ms_per_batch=round(float(slope), 4), fixed_ms=round(float(icpt), 4), mrays_per_s_k20=round(20 * (1 << 20) / row["ms"][20][0] / 1e3, 1))
        print(json.dumps(row), flush=True)
        for d in d_sets + d_outs: d.free()
        scene.destroy(); env.destroy()
