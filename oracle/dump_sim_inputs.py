"""Writes the bench scene's blobs and its 1M diffuse batch as raw files for oracle/wave_sim, oracle/split_sim and oracle/wide_sim."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rayaccel_amd import synth
from oracle import oracle as orc

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/racc_sim"
os.makedirs(out, exist_ok=True)
sc = synth.battlefield_synth()
blobs = orc.build_scene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(blobs, prim, threads=8)
diff = synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20)
blobs["nodes"].tofile(os.path.join(out, "nodes.bin"))
blobs["pairs"].tofile(os.path.join(out, "pairs.bin"))
blobs["remap"].tofile(os.path.join(out, "remap.bin"))
diff.tofile(os.path.join(out, "diffuse.bin"))
prim.tofile(os.path.join(out, "primary.bin"))
print(out, len(blobs["nodes"]), len(blobs["pairs"]), len(diff))
