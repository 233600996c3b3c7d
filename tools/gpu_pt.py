"""Device-resident path tracer at 1920x1080x64 spp (BASELINE configs[4]): Mrays/s end to end per RACC_PT_WAVES_PER_SIMD / samples per batch."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayaccel_amd import synth
from rayaccel_amd.engine import path_trace
sc = synth.battlefield_synth()
p = tempfile.mktemp(suffix=".bin")
synth.write_scene_bin(p, sc, viewport=(1920, 1080))
for w in sys.argv[1:] or ["0"]:
    if w != "0":
        os.environ["RACC_PT_WAVES_PER_SIMD"] = w
    else:
        os.environ.pop("RACC_PT_WAVES_PER_SIMD", None)
    path_trace(p, 1920, 1080, 0, 8, shading="gpu")
    best = min(path_trace(p, 1920, 1080, 0, 64, shading="gpu")[1]["seconds"] for _ in range(3))
    _, s = path_trace(p, 1920, 1080, 0, 64, shading="gpu")
    print("waves_per_simd", w, "best_s", round(best, 4), "Mrays/s", round(s["rays_traced"] / best / 1e6, 1), "rounds", s["reserved"], flush=True)
os.unlink(p)
