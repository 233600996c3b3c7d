"""PCIe-inclusive rate of the host-buffer entry point (racc_hip_intersect) on the 1M diffuse batch: pageable vs page-locked
host arrays.  RACC_SLICE=<rays> overrides the slice size of the pipelined path.
   python tools/gpu_pcie.py [rays] ['{"kernel_variant":45}']"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"])
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rays = np.ascontiguousarray(np.concatenate([synth.diffuse_bounce_rays(sc, prim, ref, 1 << 20, first_sample=s) for s in range((n + (1 << 20) - 1) >> 20)])[:n])
lib = ra.load_library()
with ra.Context(device=0, **(json.loads(sys.argv[2]) if len(sys.argv) > 2 else {})) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])
    out = np.zeros(n, ra.RESULT_DTYPE)
    def rate(reps=5):
        ctx.intersect(scene, env, rays, out)
        t = time.perf_counter()
        for _ in range(reps):
            ctx.intersect(scene, env, rays, out)
        return round(reps * n / (time.perf_counter() - t) / 1e6, 1)
    pageable = rate()
    base = out.copy()
    assert lib.racc_hip_register_host(ctx._h, rays.ctypes.data, rays.nbytes) == 0 and lib.racc_hip_register_host(ctx._h, out.ctypes.data, out.nbytes) == 0
    locked = rate()
    assert out.tobytes() == base.tobytes()
    print("rays", n, "slice", os.environ.get("RACC_SLICE", "default"), "pageable", pageable, "page-locked", locked, "Mrays/s", flush=True)
