import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from rayaccel_amd import synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("grep -m1 'model name' /proc/cpuinfo; nproc")
sc = synth.battlefield_synth()
s = orc.build_scene(sc["vertices"], sc["indices"])
rays, _ = synth.primary_rays(sc["camera"], 1024, 1024)
res = orc.traverse(s, rays, threads=16)
d = synth.diffuse_bounce_rays(sc, rays, res, 1 << 20)
out = np.zeros(len(d), orc.RESULT_DTYPE)
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    orc.traverse(s, d, threads=th, out=out)
    t = time.perf_counter(); orc.traverse(s, d, threads=th, repeat=2 if th > 1 else 1, out=out); dt = (time.perf_counter() - t) / (2 if th > 1 else 1)
    print("threads %3d: %.1f Mrays/s" % (th, len(d) / dt / 1e6), flush=True)
