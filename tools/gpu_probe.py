"""First-contact probe for the GPU box: parity vs oracle on a small + full scene, then kernel timings
for a few launch-option combinations.  Usage: python tools/gpu_probe.py [--full]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

def parity(name, got, ref):
    hit = ref["triangle"] != 0xFFFFFFFF
    prim = int((got["triangle"] != ref["triangle"]).sum())
    bits = sum(int((got[f][hit].view(np.uint32) != ref[f][hit].view(np.uint32)).sum()) for f in ("t", "u", "v"))
    col = float(np.max(np.abs(np.stack([got[f][~hit] - ref[f][~hit] for f in ("t", "u", "v")])))) if (~hit).any() else 0.0
    print("  parity[%s]: prim mismatches %d, t/u/v bit mismatches %d, max miss-colour diff %.2e, hit rate %.3f" % (name, prim, bits, col, hit.mean()), flush=True)

def main():
    full = "--full" in sys.argv
    kw = {} if full else dict(grid=150, boxes=300, quads=1000)
    t = time.time(); sc = synth.battlefield_synth(**kw); print(sc["name"], "gen %.2fs" % (time.time() - t), flush=True)
    t = time.time(); host = ra.HostScene(sc["vertices"], sc["indices"]); print("build %.2fs nodes %d pairs %d" % (time.time() - t, len(host.nodes), host.pair_count), flush=True)
    res = 1024 if full else 512
    rays, _ = synth.primary_rays(sc["camera"], res, res)
    t = time.time(); ref, nv, npp, dp = orc.traverse(host.blobs(), rays, env=sc["env"], counters=True); print("oracle primary %.2fs" % (time.time() - t), flush=True)
    bounce = synth.diffuse_bounce_rays(sc, rays, ref, len(rays))
    ref2, nv2, np2, dp2 = orc.traverse(host.blobs(), bounce, env=sc["env"], counters=True)
    B1, B2 = orc.algorithmic_bytes(ref, nv, npp), orc.algorithmic_bytes(ref2, nv2, np2)
    print("alg bytes/ray primary %.0f diffuse %.0f; nv %.1f/%.1f np %.2f/%.2f depth %d/%d" % (B1 / len(rays), B2 / len(bounce), nv.mean(), nv2.mean(), npp.mean(), np2.mean(), dp.max(), dp2.max()), flush=True)
    combos = [dict()]
    for v in (1, 10, 11, 13):
        combos.append(dict(kernel_variant=v))
    for v in (1, 10, 11):
        for w in (4, 5, 6, 8):
            combos.append(dict(kernel_variant=v, waves_per_simd=w))
    for lf, rf in ((4, 32), (8, 24), (8, 40), (8, 48), (12, 32), (16, 32), (6, 32)):
        combos.append(dict(kernel_variant=10, leaf_min=lf, refill_min=rf))
    first = True
    for opt in combos:
        with ra.Context(device=0, **opt) as ctx:
            scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
            env = ctx.create_environment(sc["env"])
            if first or ("kernel_variant" in opt and len(opt) == 1):
                print("scene info", scene.info, flush=True)
                got = ctx.intersect(scene, env, rays); parity("primary", got, ref)
                got2 = ctx.intersect(scene, env, bounce); parity("diffuse", got2, ref2)
                first = False
            out = {}
            for name, batch, B in (("primary", rays, B1), ("diffuse", bounce, B2)):
                d_r = ctx.alloc(batch.nbytes); d_o = ctx.alloc(len(batch) * 16); d_r.upload(batch)
                ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, len(batch), 3)
                ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, len(batch), 10)
                m = float(np.median(ms))
                out[name] = dict(ms=round(m, 4), mrays=round(len(batch) / m / 1e3, 1), gbs=round(B / m / 1e6, 1))
                d_r.free(); d_o.free()
            if opt.get("kernel_variant") in (9, 12):
                ctx.read_stats()
                d_r = ctx.alloc(bounce.nbytes); d_o = ctx.alloc(len(bounce) * 16); d_r.upload(bounce)
                ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, len(bounce), 1)
                st = ctx.read_stats(); d_r.free(); d_o.free()
                st["inner_util"] = round(st["inner_lanes"] / max(1, st["inner_iters"]) / 64, 3)
                st["leaf_util"] = round(st["leaf_lanes"] / max(1, st["leaf_iters"]) / 64, 3)
                out["stats_diffuse"] = st
            print(json.dumps(dict(opt=opt, launch=ctx.launch_info(), **out)), flush=True)
            scene.destroy(); env.destroy()

if __name__ == "__main__":
    main()
