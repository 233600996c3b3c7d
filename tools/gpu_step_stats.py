"""Wave-level step statistics of the default kernel from the GPU itself (kernel_variant 42 = traverseKernelV8 with counting snippets
spliced into the hot loop, racc_kernel_v8.inc): inner / leaf steps per 64 rays, live lanes per step, the share of inner steps fetched
cooperatively — and the share of inner steps in which ALL live inner lanes hold the same node, i.e. what a wave-uniform scalar step
(s_load + SGPR box operands) could serve at all (round-4 verdict, item 2).  Per batch: BASELINE configs[1] (1M coherent primaries),
configs[2] (1M first-bounce diffuse), an 8M-ray diffuse launch (closer to steady state: the drain weighs 1/8), on both trees.
    python tools/gpu_step_stats.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth

sc = synth.battlefield_synth()
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
out = {}
for quality in (1, 0):
    host = ra.HostScene(sc["vertices"], sc["indices"], quality=quality)
    with ra.Context(device=0, lanes=1, kernel_variant=42, chain_launches=0) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        hits = ctx.intersect(scene, env, prim)
        sets = synth.diffuse_bounce_batches(sc, prim, hits, 1 << 20, range(8))
        batches = (("configs[1] 1M coherent primaries", prim), ("configs[2] 1M first-bounce diffuse", sets[0]), ("8M first-bounce diffuse, one launch", np.concatenate(sets)),
                   ("64k first-bounce diffuse", sets[0][:65536]))
        for name, rays in batches:
            d_r = ctx.alloc(rays.nbytes); d_o = ctx.alloc(len(rays) * 16); d_r.upload(rays)
            ctx.read_stats(0, reset=True)
            ctx.intersect_device(scene, env, d_r.ptr, d_o.ptr, len(rays), lane=0)
            ctx.wait(0)
            st = ctx.read_stats(0, reset=True)
            raw = list(st.values())
            inner, inner_l, leaf, leaf_l, uniform, coop = raw[0], raw[1], raw[2], raw[3], raw[14], raw[15]
            per64 = 64.0 / len(rays)
            row = dict(rays=len(rays), inner_steps_per_64_rays=round(inner * per64, 2), inner_lane_occupancy=round(inner_l / max(inner, 1) / 64.0, 4),
                       leaf_steps_per_64_rays=round(leaf * per64, 2), leaf_lane_occupancy=round(leaf_l / max(leaf, 1) / 64.0, 4),
                       node_visits_per_ray=round(inner_l / len(rays), 2), pair_tests_per_ray=round(leaf_l / len(rays), 2),
                       vmem_node_and_pair_instructions_per_ray=round((4 * inner + 3 * leaf) / len(rays), 3),
                       share_of_inner_steps_wave_uniform=round(uniform / max(inner, 1), 4), share_of_inner_steps_cooperative=round(coop / max(inner, 1), 4),
                       waves=raw[7], refills_per_64_rays=round(raw[4] * per64, 2))
            out["quality %d: %s" % (quality, name)] = row
            print("quality %d: %-40s %s" % (quality, name, json.dumps(row)), flush=True)
            d_r.free(); d_o.free()
        scene.destroy(); env.destroy()
if len(sys.argv) > 1:
    json.dump(dict(what="kernel_variant 42 (statistics build of the default kernel), one launch alone on the GPU, battlefield-synth; "
                        "a step = one pass of the wave through the inner or leaf body; the DEEP door's C++ iterations (< 0.1 % of the steps) are not counted",
                   rows=out), open(sys.argv[1], "w"), indent=1)
