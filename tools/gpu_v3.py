import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc
sc = synth.battlefield_synth(grid=150, boxes=300, quads=1000)
host = ra.HostScene(sc["vertices"], sc["indices"])
rays, _ = synth.primary_rays(sc["camera"], 256, 256)
ref = orc.traverse(host.blobs(), rays, env=sc["env"], threads=8)
for v in (18, 19, 20):
    with ra.Context(device=0, kernel_variant=v) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        for n in (1, 64, 1000, 65536):
            got = ctx.intersect(scene, env, rays[:n])
            hit = ref["triangle"][:n] != 0xFFFFFFFF
            bad = int((got["triangle"] != ref["triangle"][:n]).sum()) + sum(int((got[f][hit].view(np.uint32) != ref[f][:n][hit].view(np.uint32)).sum()) for f in "tuv")
            print("variant", v, "n", n, "mismatches", bad, flush=True)
