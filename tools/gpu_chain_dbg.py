"""Debug aid for chained launches: a random sequence of device-resident batches on the engine's lanes; prints the launches whose
results differ from the oracle (ids / unshaded misses).   python tools/gpu_chain_dbg.py [launches] [seed]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc
sc = synth.battlefield_synth(grid=40, boxes=32, quads=100)
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 256, 256)
hits = orc.traverse(host.blobs(), prim)
pool = np.concatenate([prim, synth.diffuse_bounce_rays(sc, prim, hits, 50000), synth.random_rays(30011, seed=11, ymax=30.0)])
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
pool = pool[rng.permutation(len(pool))]
pool["dir"][::997] = np.nan
ref = orc.traverse(host.blobs(), pool, env=sc["env"])
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
with ra.Context(device=0) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap); env = ctx.create_environment(sc["env"])
    d_pool = ctx.alloc(pool.nbytes); d_pool.upload(pool)
    outs = []
    for k in range(launches):
        n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 20000, int(rng.integers(1, len(pool)))]))
        off = int(rng.integers(0, len(pool) - n + 1))
        d_o = ctx.alloc(n * 16)
        ctx.intersect_device(scene, env, d_pool.ptr + off * 32, d_o.ptr, n, lane=ra.LANE_AUTO)
        outs.append((d_o, off, n))
    ctx.wait(ra.LANE_AUTO)
    nbad = 0
    for i, (d_o, off, n) in enumerate(outs):
        got = d_o.download(orc.RESULT_DTYPE, n); want = ref[off:off + n]
        miss = want["triangle"] == 0xFFFFFFFF
        bad_id = int((got["triangle"] != want["triangle"]).sum())
        bad_rgb = int((~(np.abs(got["t"][miss] - want["t"][miss]) <= 1e-4)).sum())
        if bad_id or bad_rgb:
            nbad += 1
            print(i, "rays", n, "offset", off, "id mismatches", bad_id, "unshaded misses", bad_rgb, "of", int(miss.sum()), flush=True)
    print("launches", launches, "bad", nbad)
