"""Runs the REFERENCE's own OpenCL `traversal` kernel (RayAccelerator/Kernels.h:139-242) on the GPU.

TEST INFRASTRUCTURE ONLY.  oracle/_ref/traversal_gfx950.co is built by oracle/Makefile from the kernel text where it lies
in /root/reference (extracted at build time, never stored in this repo) with the image's own clang and ROCm device
libraries, using the reference's own build options (RayAccelerator.cpp:489-495: -cl-mad-enable -cl-no-signed-zeros
-cl-unsafe-math-optimizations -cl-finite-math-only -cl-fast-relaxed-math -DWORK_GROUP=8 -DWIN32=1).  This module loads that
binary through the ROCm OpenCL runtime (libOpenCL / libamdocl64, part of the image) and launches it exactly as the
reference does (RayAccelerator.cpp:378-404: global = count rounded up to 8, local = 8).  It is what pins the oracle:
the restatement in racc_oracle.c must agree with this kernel on the same buffers.

    python oracle/ref_kernel.py in.npz out.npy      (subprocess entry, keeps OpenCL out of the HIP process)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(_HERE, "_ref", "traversal_gfx950.co")

CL_DEVICE_TYPE_GPU = 4
CL_MEM_READ_WRITE, CL_MEM_READ_ONLY, CL_MEM_COPY_HOST_PTR = 1, 4, 32
CL_RGBA, CL_FLOAT, CL_MEM_OBJECT_IMAGE2D = 0x10B5, 0x10DE, 0x10F1


class _ImageFormat(C.Structure):
    _fields_ = [("order", C.c_uint32), ("type", C.c_uint32)]


class _ImageDesc(C.Structure):
    _fields_ = [("image_type", C.c_uint32), ("width", C.c_size_t), ("height", C.c_size_t), ("depth", C.c_size_t),
                ("array_size", C.c_size_t), ("row_pitch", C.c_size_t), ("slice_pitch", C.c_size_t),
                ("num_mip_levels", C.c_uint32), ("num_samples", C.c_uint32), ("buffer", C.c_void_p)]


def built():
    return os.path.exists(BINARY)


def _cl():
    for name in ("libOpenCL.so.1", "libOpenCL.so", "/opt/rocm/lib/libamdocl64.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("no OpenCL runtime")


def _run_in_process(nodes, pairs, remap, env, rays, repeats=0):
    cl = _cl()
    vp = C.c_void_p
    err = C.c_int(0)
    nplat = C.c_uint(0)
    cl.clGetPlatformIDs(0, None, C.byref(nplat))
    plats = (vp * max(1, nplat.value))()
    cl.clGetPlatformIDs(nplat.value, plats, None)
    dev = vp()
    for p in plats[: nplat.value]:
        nd = C.c_uint(0)
        if cl.clGetDeviceIDs(vp(p), C.c_ulong(CL_DEVICE_TYPE_GPU), 1, C.byref(dev), C.byref(nd)) == 0 and nd.value:
            break
    else:
        raise RuntimeError("no OpenCL GPU device")
    cl.clCreateContext.restype = vp
    ctx = vp(cl.clCreateContext(None, 1, C.byref(dev), None, None, C.byref(err)))
    assert err.value == 0, "clCreateContext %d" % err.value
    cl.clCreateCommandQueue.restype = vp
    q = vp(cl.clCreateCommandQueue(ctx, dev, C.c_ulong(0), C.byref(err)))
    blob = open(BINARY, "rb").read()
    buf = C.create_string_buffer(blob, len(blob))
    lens = (C.c_size_t * 1)(len(blob))
    ptrs = (C.c_char_p * 1)(C.cast(buf, C.c_char_p))
    status = C.c_int(0)
    cl.clCreateProgramWithBinary.restype = vp
    prog = vp(cl.clCreateProgramWithBinary(ctx, 1, C.byref(dev), lens, ptrs, C.byref(status), C.byref(err)))
    assert err.value == 0 and status.value == 0, "clCreateProgramWithBinary %d/%d" % (err.value, status.value)
    rc = cl.clBuildProgram(prog, 1, C.byref(dev), None, None, None)
    assert rc == 0, "clBuildProgram %d" % rc
    cl.clCreateKernel.restype = vp
    k = vp(cl.clCreateKernel(prog, b"traversal", C.byref(err)))
    assert err.value == 0, "clCreateKernel %d" % err.value
    cl.clCreateBuffer.restype = vp

    def mkbuf(a, flags=CL_MEM_READ_ONLY | CL_MEM_COPY_HOST_PTR):
        a = np.ascontiguousarray(a)
        b = vp(cl.clCreateBuffer(ctx, C.c_ulong(flags), C.c_size_t(a.nbytes), a.ctypes.data_as(vp), C.byref(err)))
        assert err.value == 0, "clCreateBuffer %d" % err.value
        return b, a

    n = np.ascontiguousarray(rays).nbytes // 32
    b_rays, _k1 = mkbuf(rays)
    b_nodes, _k2 = mkbuf(nodes)
    b_pairs, _k3 = mkbuf(pairs)
    b_remap, _k4 = mkbuf(remap)
    out = np.zeros((n, 4), np.float32)
    b_out = vp(cl.clCreateBuffer(ctx, C.c_ulong(CL_MEM_READ_WRITE), C.c_size_t(out.nbytes), None, C.byref(err)))
    env = np.ascontiguousarray(env, np.float32)
    fmt = _ImageFormat(CL_RGBA, CL_FLOAT)
    desc = _ImageDesc(CL_MEM_OBJECT_IMAGE2D, env.shape[1], env.shape[0], 0, 0, env.shape[1] * 16, 0, 0, 0, None)   # Environment.cpp:36-48
    cl.clCreateImage.restype = vp
    img = vp(cl.clCreateImage(ctx, C.c_ulong(CL_MEM_READ_ONLY | CL_MEM_COPY_HOST_PTR), C.byref(fmt), C.byref(desc), env.ctypes.data_as(vp), C.byref(err)))
    if err.value != 0:
        # CL_INVALID_OPERATION (-59) on MI355X: CDNA has no image/sampler hardware (CL_DEVICE_IMAGE_SUPPORT = 0) and the
        # compiled kernel contains no image instruction at all — the reference's miss colour cannot be produced on this
        # device.  Pass a null image: the hit side (primId, t, u, v — the parity criterion) does not touch it.
        img = vp(None)
        print("note: no OpenCL image support on this device (clCreateImage %d); miss colours are not evaluated" % err.value, file=sys.stderr)
    count = C.c_int(n)
    for i, (sz, ref) in enumerate(((8, b_rays), (8, b_nodes), (8, b_pairs), (8, b_remap), (8, b_out), (4, count), (8, img))):   # RayAccelerator.cpp:383-399
        rc = cl.clSetKernelArg(k, i, C.c_size_t(sz), C.byref(ref))
        assert rc == 0, "clSetKernelArg(%d) %d" % (i, rc)
    gsz = (C.c_size_t * 1)((n + 7) & ~7)          # RayAccelerator.cpp:380-381
    lsz = (C.c_size_t * 1)(8)
    rc = cl.clEnqueueNDRangeKernel(q, k, 1, None, gsz, lsz, 0, None, None)
    assert rc == 0, "clEnqueueNDRangeKernel %d" % rc
    rc = cl.clFinish(q)
    assert rc == 0, "clFinish %d" % rc
    rc = cl.clEnqueueReadBuffer(q, b_out, 1, C.c_size_t(0), C.c_size_t(out.nbytes), out.ctypes.data_as(vp), 0, None, None)
    assert rc == 0, "clEnqueueReadBuffer %d" % rc
    times = []
    import time
    for _ in range(repeats):                     # enqueue + clFinish per launch, as the reference's GPU thread does
        t0 = time.perf_counter()
        cl.clEnqueueNDRangeKernel(q, k, 1, None, gsz, lsz, 0, None, None)
        cl.clFinish(q)
        times.append(time.perf_counter() - t0)
    return out, times


def run(scene, rays, env, repeats=0):
    """scene: dict(nodes, pairs, remap) in the reference format; returns RESULT-layout [N] records (and, with repeats > 0,
    the wall time of each extra enqueue + clFinish in seconds)."""
    import tempfile
    from . import oracle as orc
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.npz"), os.path.join(d, "out.npy")
        np.savez(fin, nodes=scene["nodes"].view(np.uint8), pairs=scene["pairs"].view(np.uint8), remap=scene["remap"],
                 env=np.ascontiguousarray(env, np.float32), rays=np.ascontiguousarray(rays).view(np.uint8))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), fin, fout, str(repeats)], capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            raise RuntimeError("reference kernel run failed: " + p.stdout[-2000:] + p.stderr[-2000:])
        res = np.load(fout).view(orc.RESULT_DTYPE).reshape(-1)
        if repeats:
            return res, np.load(fout + ".times.npy")
        return res


if __name__ == "__main__":
    z = np.load(sys.argv[1])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    res, times = _run_in_process(z["nodes"], z["pairs"], z["remap"], z["env"], z["rays"], reps)
    np.save(sys.argv[2], res)
    np.save(sys.argv[2] + ".times.npy", np.array(times))
