import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc
sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
rays, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref0 = orc.traverse(host.blobs(), rays, threads=16)
big = synth.diffuse_bounce_rays(sc, rays, ref0, 1 << 20)
ref = orc.traverse(host.blobs(), big, env=sc["env"], threads=16)
n = len(big)
with ra.Context(device=0) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])
    d_r = ctx.alloc(big.nbytes); d_r.upload(big)
    outs = [ctx.alloc(n * 16) for _ in range(4)]
    for rep in range(3):
        for k in range(8):
            ctx.intersect_device(scene, env, d_r.ptr, outs[k % 4].ptr, n, lane=k % 4)
        for lane in range(4):
            ctx.wait(lane)
        for lane in range(4):
            got = outs[lane].download(ra.RESULT_DTYPE, n)
            bad = np.nonzero(got["triangle"] != ref["triangle"])[0]
            hit = ref["triangle"] != 0xFFFFFFFF
            badt = np.nonzero(got["t"][hit].view(np.uint32) != ref["t"][hit].view(np.uint32))[0]
            miss = ~hit
            badc = np.nonzero(np.abs(got["t"][miss] - ref["t"][miss]) > 1e-4)[0]
            print("rep", rep, "lane", lane, "prim mismatches", len(bad), "t mismatches", len(badt), "miss colour mismatches", len(badc), bad[:5], flush=True)
            if len(badc):
                i = np.nonzero(miss)[0][badc[:3]]
                print("   e.g.", got[i], ref[i])
