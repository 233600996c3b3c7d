import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc
sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
diff = synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20)
big = np.tile(diff, 40)          # 41.9M rays
n = len(big)
with ra.Context(device=0) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])
    d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(big)
    ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 3)
    out = d_o.download(orc.RESULT_DTYPE, n)
    one = orc.traverse(host.blobs(), diff, env=sc["env"], threads=16)
    ok = all(np.array_equal(out[k * (1 << 20):(k + 1) * (1 << 20)]["triangle"], one["triangle"]) for k in (0, 17, 39))
    print("rays", n, "ms", [round(m, 3) for m in ms], "Grays/s", round(n / min(ms) / 1e6, 2), "primIds ok", ok)
