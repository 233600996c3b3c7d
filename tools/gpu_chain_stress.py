"""Repeats the launch sequence of tests/test_gpu_parity.py::test_chained_launches (chained launches of very different sizes over the
lanes, two scenes, ring laps) and, when a batch differs from the oracle, says where: launch index, size, the indices that differ, what
the records hold.   python tools/gpu_chain_stress.py [rounds] [seed]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sc = synth.battlefield_synth(grid=40, boxes=32, quads=100)
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 256, 256)
hits = orc.traverse(host.blobs(), prim)
rng = np.random.default_rng(seed)
pool = np.concatenate([prim, synth.diffuse_bounce_rays(sc, prim, hits, 50000), synth.random_rays(30011, seed=11, ymax=30.0)])
pool = pool[rng.permutation(len(pool))]
pool["dir"][::997] = np.nan
ref = orc.traverse(host.blobs(), pool, env=sc["env"])
other = synth.battlefield_synth(grid=24, boxes=8, quads=30)
other_host = ra.HostScene(other["vertices"], other["indices"])
ref_other = orc.traverse(other_host.blobs(), pool[:5000], env=sc["env"])
bad_rounds = 0
for rnd in range(rounds):
    with ra.Context(device=0, chain_min_rays=1) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        scene2 = ctx.upload_scene(other_host.nodes, other_host.pairs, other_host.remap)
        env = ctx.create_environment(sc["env"])
        d_pool = ctx.alloc(pool.nbytes); d_pool.upload(pool)
        issued = []
        for k in range(300):
            n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 20000, int(rng.integers(1, len(pool)))]))
            if os.environ.get("RACC_STRESS_FIRST") and k < 2:      # a fresh context's first two launches: a long one, a short one chained behind it
                n = (int(os.environ["RACC_STRESS_FIRST"]), 20000)[k]
            off = int(rng.integers(0, len(pool) - n + 1))
            d_o = ctx.alloc(n * 16)
            fill = np.full(n * 4, 0xABABABAB, np.uint32)      # so that "never written" is recognisable
            d_o.upload(fill)
            if k % 37 == 36:
                m = min(n, 5000)
                ctx.intersect_device(scene2, env, d_pool.ptr, d_o.ptr, m, lane=ra.LANE_AUTO); issued.append((d_o, 0, m, ref_other))
            else:
                ctx.intersect_device(scene, env, d_pool.ptr + off * 32, d_o.ptr, n, lane=ra.LANE_AUTO); issued.append((d_o, off, n, ref))
            if k % 53 == 52:
                ctx.wait(ra.LANE_AUTO)
        ctx.wait(ra.LANE_AUTO)
        nbad = 0
        for i, (d_o, off, n, want) in enumerate(issued):
            got = d_o.download(orc.RESULT_DTYPE, n)
            w = want[off:off + n]
            hit = w["triangle"] != 0xFFFFFFFF
            diff = (got["triangle"] != w["triangle"]) | (hit & ((got["t"].view(np.uint32) != w["t"].view(np.uint32)))) | (~hit & ~np.isclose(got["t"], w["t"], rtol=1e-5, atol=1e-5))
            if diff.any():
                idx = np.nonzero(diff)[0]
                raw = got.view(np.uint32).reshape(-1, 4)
                never = int((raw[idx] == 0xABABABAB).all(1).sum())
                parked = int(((got["triangle"][idx] == 0xFFFFFFFF) & (np.abs(got["t"][idx] ** 2 + got["u"][idx] ** 2 + got["v"][idx] ** 2 - 1.0) < 1e-3)).sum())
                print(json.dumps(dict(round=rnd, launch=i, rays=n, lane=i % 3, differing=len(idx), first=int(idx[0]), last=int(idx[-1]), never_written=never, parked_direction=parked,
                                      prev_rays=[issued[j][2] for j in range(max(0, i - 3), i)])), flush=True)
                nbad += 1
            d_o.free()
        bad_rounds += nbad > 0
        d_pool.free(); scene.destroy(); scene2.destroy(); env.destroy()
print("rounds %d, rounds with a differing batch: %d" % (rounds, bad_rounds))
