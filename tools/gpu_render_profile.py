import os, sys, tempfile
sys.path.insert(0, "/root/repo")
os.environ["RACC_PROFILE"] = "1"
from rayaccel_amd import synth
from rayaccel_amd.engine import path_trace
sc = synth.battlefield_synth()
tmp = tempfile.NamedTemporaryFile(suffix=".bin", delete=False); tmp.close()
synth.write_scene_bin(tmp.name, sc, viewport=(1920, 1080))
for th in (16, 14, 8):
    _, st = path_trace(tmp.name, 1920, 1080, 0, 8, device=0, shading="cpu", cpu_threads=th)
    print(th, round(st["rays_traced"] / st["seconds"] / 1e6, 1), "Mrays/s", flush=True)
