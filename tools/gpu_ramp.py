import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc
sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
rays, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), rays, threads=16)
big = synth.diffuse_bounce_rays(sc, rays, ref, 1 << 20)
with ra.Context(device=0, kernel_variant=1) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])
    d_r = ctx.alloc(big.nbytes); d_o = ctx.alloc(len(big) * 16); d_r.upload(big)
    t0 = time.time()
    ms = ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, len(big), 2000)
    wall = time.time() - t0
    ms = np.array(ms)
    print("wall %.3f s for 2000 launches; sum kernel %.3f s" % (wall, ms.sum() / 1e3))
    for a, b in ((0, 10), (10, 50), (50, 100), (100, 200), (200, 500), (500, 1000), (1000, 2000)):
        print("launch %4d-%4d: median %.4f ms min %.4f" % (a, b, np.median(ms[a:b]), ms[a:b].min()))
