"""How much of the fixed (drain) cost depends on which rays come last?  Same 1M diffuse batch, different orders."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
diff = synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20)
out, nv, npp, dp = orc.traverse(host.blobs(), diff, counters=True) if len(sys.argv) > 1 else (None, None, None, None)
n = len(diff)
rng = np.random.default_rng(1)
orders = dict(forward=np.arange(n), reverse=np.arange(n)[::-1].copy(),
              blocks64_shuffled=(rng.permutation(n // 64)[:, None] * 64 + np.arange(64)[None, :]).reshape(-1),
              blocks4096_shuffled=(rng.permutation(n // 4096)[:, None] * 4096 + np.arange(4096)[None, :]).reshape(-1),
              fully_shuffled=rng.permutation(n))
if nv is not None:
    cost = nv.astype(np.int64) + 2 * npp
    orders["expensive_first"] = np.argsort(-cost, kind="stable")
    orders["cheap_first"] = np.argsort(cost, kind="stable")
    # coherent order kept, but the 5 % most expensive rays moved to the front
    k = n // 20
    top = np.argsort(-cost, kind="stable")[:k]
    mask = np.ones(n, bool); mask[top] = False
    orders["top5pct_first"] = np.concatenate([np.sort(top), np.nonzero(mask)[0]])
with ra.Context(device=0) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])
    d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16)
    for name, o in orders.items():
        d_r.upload(np.ascontiguousarray(diff[o]))
        ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 3)
        ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 20)
        print(json.dumps(dict(order=name, ms=round(float(np.median(ms)), 4))), flush=True)
