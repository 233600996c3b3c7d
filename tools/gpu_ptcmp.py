import sys, os, tempfile, numpy as np
sys.path.insert(0, os.getcwd())
from rayaccel_amd import synth
from rayaccel_amd.engine import path_trace
sc = synth.battlefield_synth()
p = tempfile.mktemp(suffix=".bin")
synth.write_scene_bin(p, sc, viewport=(1920, 1080))
g, sg = path_trace(p, 1920, 1080, 0, 4, shading="gpu")
c, scpu = path_trace(p, 1920, 1080, 0, 4, shading="cpu")
print("equal", np.array_equal(g, c), sg["rays_traced"], scpu["rays_traced"], sg["seconds"], scpu["seconds"])
for b in (1, 2, 4, 8, 16):
    g2, s2 = path_trace(p, 1920, 1080, 0, 64, shading="gpu", samples_per_batch=b)
    print("batch", b, round(s2["rays_traced"] / s2["seconds"] / 1e6, 1), "Mrays/s", round(s2["seconds"], 4), "s rounds", s2["reserved"])
os.unlink(p)
