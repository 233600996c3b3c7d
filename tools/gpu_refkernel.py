import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc, ref_kernel
from rayaccel_amd import synth
os.system("clinfo 2>/dev/null | grep -E 'Platform Name|Device Name|Number of devices' | head -5")
print("binary built:", ref_kernel.built())
sc = synth.battlefield_synth(grid=40, boxes=32, quads=100)
s = orc.build_scene(sc["vertices"], sc["indices"])
rays, _ = synth.primary_rays(sc["camera"], 128, 128)
ref = orc.traverse(s, rays, env=sc["env"])
for a in sys.argv[1:]:
    if a.startswith("--n="):
        rays = rays[:int(a[4:])]; ref = ref[:len(rays)]
if "--hits-only" in sys.argv:
    rays = rays[ref["triangle"] != 0xFFFFFFFF]
    ref = orc.traverse(s, rays, env=sc["env"])
got = ref_kernel.run(s, rays, sc["env"])
hit = ref["triangle"] != 0xFFFFFFFF
print("hit/miss disagreements", int(((got["triangle"] != 0xFFFFFFFF) != hit).sum()), "of", len(rays))
both = hit & (got["triangle"] != 0xFFFFFFFF)
print("primId mismatches", int((got["triangle"][both] != ref["triangle"][both]).sum()))
same = both & (got["triangle"] == ref["triangle"])
for f in "tuv":
    d = np.abs(got[f][same] - ref[f][same]); print(f, "max abs", d.max(), "max rel", (d / np.maximum(np.abs(ref[f][same]), 1e-6)).max())
m = ~hit & (got["triangle"] == 0xFFFFFFFF)
print("miss colour max abs diff", np.abs(np.stack([got[f][m] - ref[f][m] for f in "tuv"])).max(), "range", ref["t"][m].max())
