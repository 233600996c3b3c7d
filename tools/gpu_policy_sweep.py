"""Scheduling-policy sweep on the two figures that matter for a caller issuing batch after batch: the 4M-ray launch (steady
   state) and 1M-ray launches back to back over the lanes.   python tools/gpu_policy_sweep.py '{"refill_min":20}' ...
   RACC_SWEEP_SCENE=city-synth|soup-synth: the same sweep on another scene class (round 6: were the constants tuned to one scene family?).
   RACC_SWEEP_SCENE=battlefield-synth-xl: Scheduling-policy sweep on battlefield-synth-XL with its 1M incoherent rays (where the fabric binds): one launch alone and 40 chained.
   RACC_SWEEP_SCENE=battlefield-synth-xl python tools/gpu_policy_sweep.py '{"refill_min":20}' ..."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc


def sweep():
    sc = synth.SCENES[os.environ.get("RACC_SWEEP_SCENE", "battlefield-synth")]()
    host = ra.HostScene(sc["vertices"], sc["indices"], quality=int(os.environ.get("RACC_SWEEP_QUALITY", "1")))
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    ref = orc.traverse(host.blobs(), prim, threads=16)
    diff = np.concatenate(synth.diffuse_bounce_batches(sc, prim, ref, 1 << 20, range(4)))
    combos = [json.loads(a) for a in sys.argv[1:]] or [dict()] + [dict(leaf_min=l, refill_min=r, inner_reps=i) for l in (6, 12, 20) for r in (20, 32, 44) for i in (2, 3, 5)] + [dict(tail_active=t) for t in (16, 24, 40, 48)] + [dict(chunk=c) for c in (32, 128, 256)]
    for opt in combos:
        with ra.Context(device=0, lanes=4, **opt) as ctx:
            scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
            env = ctx.create_environment(sc["env"])
            n = 1 << 22
            d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(diff)
            ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 2)
            ms4 = float(np.median(ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 8)))
            n1 = 1 << 20
            outs = [ctx.alloc(n1 * 16) for _ in range(3)]
            for k in range(6): ctx.intersect_device(scene, env, d_r.ptr, outs[k % 3].ptr, n1, lane=ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for k in range(40): ctx.intersect_device(scene, env, d_r.ptr + (k % 4) * n1 * 32, outs[k % 3].ptr, n1, lane=ra.LANE_AUTO)
                ctx.wait(ra.LANE_AUTO)
                best = min(best, (time.perf_counter() - t0) / 40)
            small = {}
            for nn in (1 << 16, 1 << 20):
                ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, nn, 2)
                small[nn] = round(float(np.median(ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, nn, 10))), 4)
            print(json.dumps(dict(opt=opt, ms_4M=round(ms4, 4), overlapped_ms=round(best * 1e3, 4), ms_64k=small[1 << 16], ms_1M=small[1 << 20])), flush=True)
            scene.destroy(); env.destroy(); d_r.free(); d_o.free(); [o.free() for o in outs]


def sweep_xl():
    sc = synth.battlefield_synth_xl()
    host = ra.HostScene(sc["vertices"], sc["indices"], quality=int(os.environ.get("RACC_SWEEP_QUALITY", "1")))
    rays = synth.random_rays(1 << 20, 7)
    base = None
    for arg in sys.argv[1:] or ["{}"]:
        opt = json.loads(arg)
        with ra.Context(device=0, **opt) as ctx:
            scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
            n = len(rays)
            d_r = ctx.alloc(n * 32); d_r.upload(rays)
            outs = [ctx.alloc(n * 16) for _ in range(4)]
            ctx.intersect_device_timed(scene, None, d_r.ptr, outs[0].ptr, n, 4)
            ms = float(np.median(ctx.intersect_device_timed(scene, None, d_r.ptr, outs[0].ptr, n, 10)))
            got = outs[0].download(ra.RESULT_DTYPE, n).tobytes()
            if base is None: base = got
            for k in range(8): ctx.intersect_device(scene, None, d_r.ptr, outs[k % 4].ptr, n, lane=ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
            t = time.perf_counter()
            for k in range(40):
                ctx.intersect_device(scene, None, d_r.ptr, outs[k % 4].ptr, n, lane=ra.LANE_AUTO)
                if k % 4 == 3: ctx.wait(ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
            dt = (time.perf_counter() - t) / 40
            print(json.dumps(dict(opt=opt, ms_alone=round(ms, 4), back_to_back_ms=round(dt * 1e3, 4), same_results=(got == base))), flush=True)
            scene.destroy(); d_r.free(); [o.free() for o in outs]


if __name__ == "__main__":
    if os.environ.get("RACC_SWEEP_SCENE") == "battlefield-synth-xl": sweep_xl()
    else: sweep()
