"""A/B of the LDS node cache (kernel variants 60-63) against the default kernel at 5 and 4 waves per SIMD, on the quality tree:
   one launch at a time (1M primary, 1M / 4M / 64k diffuse), 20 and 100 chained 1M-ray batches of four rotating sample sets.  Every variant's
   results must be byte-identical to the first one's (the default kernel, which the suite holds to the oracle).
   python tools/gpu_cache_ab.py [--quality 1] [--xl] '{}' '{"waves_per_simd": 4}' 60 61 62 63"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

args = sys.argv[1:]
quality = 1
if args and args[0] == "--quality":
    quality = int(args[1]); args = args[2:]
sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=quality)
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
sets = [synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20, first_sample=s) for s in range(4)]
diff = np.concatenate(sets)
base = {}


def same(out, ref_bytes, v, name, row):
    """Byte-identical to the first variant's records — except for the 4-wide kernels (kernel_variant 45-53), which may differ on exact-distance
    ties and arbiter-confirmed closer hits (tests/helpers.py::assert_same_closest_hit holds them to that in the suite): there the number of
    differing records is reported and bounded."""
    if out.tobytes() == ref_bytes:
        return
    kv = v.get("kernel_variant", 0) if isinstance(v, dict) else v
    assert 45 <= kv <= 53, "%s differs on %s" % (v, name)
    ref = np.frombuffer(ref_bytes, orc.RESULT_DTYPE)
    n = int((out.view(np.uint8).reshape(-1, 16) != ref.view(np.uint8).reshape(-1, 16)).any(1).sum())
    hit = ref["triangle"] != 0xFFFFFFFF
    n_hit = int(((out.view(np.uint8).reshape(-1, 16) != ref.view(np.uint8).reshape(-1, 16)).any(1) & hit).sum())
    row.setdefault("records_differing_from_first", {})[name] = [n_hit, n - n_hit]      # [hit records, miss records (colours to rounding)]
    # (ties, closer hits, and — in a tree with spatial splits — the same triangle through another of its references: up to 0.5 %)
    assert n_hit <= max(8, len(ref) // 200), "%s: %d hit records differ on %s" % (v, n_hit, name)


variants = [json.loads(v) for v in args] or [{}, {"waves_per_simd": 4}, 60, 61, 62, 63]      # an int = kernel_variant, a dict = Context options
for v in variants:
    with ra.Context(device=0, **(v if isinstance(v, dict) else dict(kernel_variant=v))) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        row = dict(opt=v)
        for name, rays, n in (("primary_1M", prim, 1 << 20), ("diffuse_1M", diff, 1 << 20), ("diffuse_4M", diff, 1 << 22), ("diffuse_64K", diff, 1 << 16)):
            d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(rays[:n])
            ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 3)
            ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 20)
            row[name] = round(float(np.median(ms)), 4)
            out = d_o.download(orc.RESULT_DTYPE, n)
            if name in base:
                same(out, base[name], v, name, row)
            else:
                base[name] = out.tobytes()
            d_r.free(); d_o.free()
        # batches issued back to back over the lanes (what bench.py times): chained for the kernels that have a chained instantiation
        d_sets = []
        for s in sets:
            d = ctx.alloc(s.nbytes); d.upload(s); d_sets.append(d)
        d_outs = [ctx.alloc((1 << 20) * 16) for _ in range(8)]
        for steps in (20, 100):
            best = 1e9
            for rep in range(4):
                ctx.synchronize()
                t0 = time.perf_counter()
                for k in range(steps):
                    ctx.intersect_device(scene, env, d_sets[k % 4].ptr, d_outs[k % 8].ptr, 1 << 20, lane=ra.LANE_AUTO)
                ctx.wait(ra.LANE_AUTO)
                ctx.synchronize()
                best = min(best, (time.perf_counter() - t0) / steps * 1e3)
            row["chained_%d_ms_per_step" % steps] = round(best, 4)
            row["chained_%d_mrays_per_s" % steps] = round((1 << 20) / best / 1e3, 1)
        for k in range(4):
            out = d_outs[(96 + k) % 8].download(orc.RESULT_DTYPE, 1 << 20)
            key = "set%d" % k
            if key in base:
                same(out, base[key], v, "chained " + key, row)
            else:
                base[key] = out.tobytes()
        info = ctx.launch_info(0) if hasattr(ctx, "launch_info") else None
        if info: row["launch"] = info
        print(json.dumps(row), flush=True)
        for d in d_sets + d_outs: d.free()
        scene.destroy(); env.destroy()
