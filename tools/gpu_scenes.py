"""Three scene classes under the kernel's DEFAULT policy constants (round-5 verdict, item 8): battlefield-synth (the stand-in every BASELINE
config is measured on), city-synth (axis-aligned, areas over seven decades) and soup-synth (unconnected overlapping triangles, leaves of up to
13 pairs, stacks 20-27 deep).  Per scene and tree (the reference builder's = quality 0, the library default = quality 1): every record of the 1M
first-bounce diffuse batch and of the 1M coherent primaries against the oracle (bit-exact, or the tool exits), node visits / pair tests per ray,
one launch at a time and 20 lazily chained batches; kernel_variant 50 (fast mode) beside it.   python tools/gpu_scenes.py > profiles/<round>/scenes.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

MISS = 0xFFFFFFFF
THREADS = int(os.environ.get("RACC_TOOL_THREADS", "16"))


def exact(got, ref):
    hit = ref["triangle"] != MISS
    return np.array_equal(got["triangle"], ref["triangle"]) and all(np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)) for f in ("t", "u", "v"))


for name, make in synth.SCENES.items():
    sc = make()
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    for quality in (0, 1):
        t0 = time.perf_counter()
        host = ra.HostScene(sc["vertices"], sc["indices"], quality=quality)
        build_s = time.perf_counter() - t0
        blobs = host.blobs()
        ref_p, nvp, npp, dpp = orc.traverse(blobs, prim, env=sc["env"], counters=True, threads=THREADS)
        sets = synth.diffuse_bounce_batches(sc, prim, ref_p, 1 << 20, range(4))
        ref_d, nvd, npd, dpd = orc.traverse(blobs, sets[0], env=sc["env"], counters=True, threads=THREADS)
        row = dict(scene=sc["name"], triangles=len(sc["indices"]), quality=quality, build_seconds=round(build_s, 2), inner_nodes=len(host.nodes), pairs=int(host.pair_count),
                   primary=dict(hit_rate=round(float((ref_p["triangle"] != MISS).mean()), 3), node_visits_per_ray=round(float(nvp.mean()), 2), pair_tests_per_ray=round(float(npp.mean()), 2), max_stack=int(dpp.max())),
                   diffuse=dict(hit_rate=round(float((ref_d["triangle"] != MISS).mean()), 3), node_visits_per_ray=round(float(nvd.mean()), 2), pair_tests_per_ray=round(float(npd.mean()), 2), max_stack=int(dpd.max()),
                                algorithmic_bytes=int(orc.algorithmic_bytes(ref_d, nvd, npd))))
        for label, opts in (("default_kernel", {}), ("kernel_variant_50", dict(kernel_variant=50))):
            with ra.Context(device=0, **opts) as ctx:
                scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
                env = ctx.create_environment(sc["env"])
                row.setdefault("max_leaf_pairs", scene.info["max_leaf_pairs"]); row.setdefault("inner_height", scene.info["inner_height"])
                res = {}
                for key, rays, ref in (("primary", prim, ref_p), ("diffuse", sets[0], ref_d)):
                    d_r = ctx.alloc(rays.nbytes); d_o = ctx.alloc(len(rays) * 16); d_r.upload(rays)
                    ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, len(rays), 3)
                    ms = float(np.median(ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, len(rays), 15)))
                    got = d_o.download(orc.RESULT_DTYPE, len(rays))
                    if label == "default_kernel":
                        if not exact(got, ref):
                            sys.exit("%s quality %d %s: the default kernel's records differ from the oracle's" % (name, quality, key))
                        res[key + "_bit_exact_vs_oracle"] = True
                    else:
                        hit = ref["triangle"] != MISS
                        res[key + "_hit_records_differing_from_oracle"] = int(((got["triangle"] != ref["triangle"]) | (got["t"].view(np.uint32) != ref["t"].view(np.uint32)))[hit].sum() + ((got["triangle"] != MISS) != hit).sum())
                    res[key + "_1M_ms"] = round(ms, 4); res[key + "_1M_mrays_per_s"] = round(len(rays) / ms / 1e3, 1)
                    d_r.free(); d_o.free()
                d_sets = []
                for s in sets:
                    d = ctx.alloc(s.nbytes); d.upload(s); d_sets.append(d)
                outs = [ctx.alloc((1 << 20) * 16) for _ in range(20)]
                best = 1e9
                for rep in range(4):
                    ctx.synchronize(); t1 = time.perf_counter()
                    for k in range(20):
                        ctx.intersect_device(scene, env, d_sets[k % 4].ptr, outs[k].ptr, 1 << 20, lane=ra.LANE_AUTO)
                    ctx.wait(ra.LANE_AUTO); ctx.synchronize()
                    best = min(best, time.perf_counter() - t1)
                if label == "default_kernel" and not exact(outs[16].download(orc.RESULT_DTYPE, 1 << 20), ref_d):
                    sys.exit("%s quality %d: chained batch differs from the oracle" % (name, quality))
                res["diffuse_20_chained_mrays_per_s"] = round(20 * (1 << 20) / best / 1e6, 1)
                row[label] = res
                for d in d_sets + outs: d.free()
                scene.destroy(); env.destroy()
        print(json.dumps(row), flush=True)
