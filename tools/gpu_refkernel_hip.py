"""Cross-check: launch oracle/_ref/traversal_gfx950.co through the HIP module API (explicit kernel params)."""
import ctypes as C, os, sys, faulthandler; faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc, ref_kernel
from rayaccel_amd import synth
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
vp = C.c_void_p
def ck(rc, what):
    assert rc == 0, "%s -> %d" % (what, rc)
sc = synth.battlefield_synth(grid=40, boxes=32, quads=100)
s = orc.build_scene(sc["vertices"], sc["indices"])
rays, _ = synth.primary_rays(sc["camera"], 128, 128)
n = int(sys.argv[1]) if len(sys.argv) > 1 else len(rays)
rays = rays[:n]
ref = orc.traverse(s, rays, env=sc["env"])
blob = open(ref_kernel.BINARY, "rb").read()
mod, fn = vp(), vp()
ck(hip.hipInit(0), "hipInit"); ck(hip.hipSetDevice(0), "hipSetDevice")
ck(hip.hipModuleLoadData(C.byref(mod), blob), "hipModuleLoadData")
ck(hip.hipModuleGetFunction(C.byref(fn), mod, b"traversal"), "hipModuleGetFunction")
def dev(a):
    a = np.ascontiguousarray(a); p = vp()
    ck(hip.hipMalloc(C.byref(p), C.c_size_t(max(a.nbytes, 64))), "hipMalloc")
    ck(hip.hipMemcpy(p, a.ctypes.data_as(vp), C.c_size_t(a.nbytes), 1), "H2D")
    return p
d_rays, d_nodes, d_pairs, d_remap = dev(rays), dev(s["nodes"]), dev(s["pairs"]), dev(s["remap"])
d_out = dev(np.zeros((n, 4), np.float32)); d_img = dev(np.zeros(64, np.uint32))
count = C.c_int(n)
import struct
karg = C.create_string_buffer(struct.pack("<QQQQQiiQ", d_rays.value, d_nodes.value, d_pairs.value, d_remap.value, d_out.value, n, 0, d_img.value))
ksize = C.c_size_t(56)
extra = (vp * 5)(1, C.addressof(karg), 2, C.addressof(ksize), 3)     # HIP_LAUNCH_PARAM_BUFFER_POINTER / _SIZE / _END
blocks = (n + 7) // 8
ck(hip.hipModuleLaunchKernel(fn, blocks, 1, 1, 8, 1, 1, 0, None, None, extra), "launch")
ck(hip.hipDeviceSynchronize(), "sync")
out = np.zeros((n, 4), np.float32)
ck(hip.hipMemcpy(out.ctypes.data_as(vp), d_out, C.c_size_t(out.nbytes), 2), "D2H")
got = out.view(orc.RESULT_DTYPE).reshape(-1)
hit = ref["triangle"] != 0xFFFFFFFF
print("n", n, "hit/miss disagreements", int(((got["triangle"] != 0xFFFFFFFF) != hit).sum()))
both = hit & (got["triangle"] != 0xFFFFFFFF)
print("primId mismatches", int((got["triangle"][both] != ref["triangle"][both]).sum()), "of", int(both.sum()))
same = both & (got["triangle"] == ref["triangle"])
for f in "tuv":
    d = np.abs(got[f][same] - ref[f][same]); print(f, "max abs", d.max(), "max rel", (d / np.maximum(np.abs(ref[f][same]), 1e-6)).max())
