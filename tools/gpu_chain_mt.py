import os, sys, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc
sc = synth.battlefield_synth(grid=40, boxes=32, quads=100)
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 256, 256)
hits = orc.traverse(host.blobs(), prim)
pool = np.concatenate([prim, synth.diffuse_bounce_rays(sc, prim, hits, 50000)])
ref = orc.traverse(host.blobs(), pool, env=sc["env"])
with ra.Context(device=0) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap); env = ctx.create_environment(sc["env"])
    d_pool = ctx.alloc(pool.nbytes); d_pool.upload(pool)
    bad = []
    def work(seed):
        rng = np.random.default_rng(seed)
        for rnd in range(6):
            outs = []
            for k in range(60):
                n = int(rng.choice([1, 64, 1000, 20000, int(rng.integers(1, len(pool)))]))
                off = int(rng.integers(0, len(pool) - n + 1))
                d_o = ctx.alloc(n * 16)
                ctx.intersect_device(scene, env, d_pool.ptr + off * 32, d_o.ptr, n, lane=ra.LANE_AUTO)
                outs.append((d_o, off, n))
            ctx.wait(ra.LANE_AUTO)
            for d_o, off, n in outs:
                got = d_o.download(orc.RESULT_DTYPE, n); want = ref[off:off + n]
                miss = want["triangle"] == 0xFFFFFFFF
                if (got["triangle"] != want["triangle"]).any() or (~(np.abs(got["t"][miss] - want["t"][miss]) <= 1e-4)).any():
                    bad.append((seed, rnd, n))
                d_o.free()
    ts = [threading.Thread(target=work, args=(s,)) for s in (1, 2, 3)]
    for t in ts: t.start()
    for t in ts: t.join()
    print("threads 3 x 360 launches, bad:", len(bad), bad[:5])
