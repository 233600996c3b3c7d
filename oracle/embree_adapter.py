"""Optional system-Embree adapter: the reference's CPU path IS Embree (RayAccelerator/Scene.cpp:374-484),
a binary-only dependency absent from the reference checkout and from both boxes of this build.  If the machine this runs
on has a system Embree 3.x/4.x (`ldconfig` / ctypes.util.find_library), this module binds it through oracle/embree_shim.c
(dlopen at run time: never a build dependency) and offers

    available()                         -> bool
    trace(scene_dict, rays, threads)    -> Result array (primId, t, u, v; misses zeroed)   second-opinion parity check
    time_batch(scene_dict, rays, thr)   -> dict for bench.py's cpu_baseline["embree"] row  a true "CPU + Embree" Mrays/s

TEST INFRASTRUCTURE ONLY; nothing under rayaccel_amd/ imports it.  With no Embree installed (the case everywhere so far)
available() is False, tests/test_gpu_embree.py skips and bench.py's row is absent."""
import ctypes as C
import ctypes.util
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SHIM_SRC = os.path.join(_HERE, "embree_shim.c")
_SHIM_LIB = os.path.join(_HERE, "libracc_embree_shim.so")
RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("minT", "<f4"), ("dir", "<f4", 3), ("maxT", "<f4")])
RESULT_DTYPE = np.dtype([("triangle", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])


def find_library():
    """Path/soname of a system Embree 4 or 3, or None.  RACC_EMBREE_LIB overrides."""
    if os.environ.get("RACC_EMBREE_LIB"):
        return os.environ["RACC_EMBREE_LIB"]
    for name in ("embree4", "embree3", "embree"):
        found = ctypes.util.find_library(name)
        if found:
            return found
    return None


def build_shim(force=False):
    if force or not os.path.exists(_SHIM_LIB) or os.path.getmtime(_SHIM_SRC) > os.path.getmtime(_SHIM_LIB):
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-shared", "-fPIC", "-Wall", _SHIM_SRC, "-o", _SHIM_LIB, "-ldl", "-lpthread"])
    lib = C.CDLL(_SHIM_LIB)
    lib.shim_open.restype = C.c_void_p
    lib.shim_open.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    lib.shim_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.shim_close.argtypes = [C.c_void_p]
    lib.shim_version.argtypes = [C.c_void_p]
    return lib


def available():
    return find_library() is not None


class Scene:
    def __init__(self, vertices, indices):
        self._lib = build_shim()
        v = np.ascontiguousarray(vertices, np.float32)
        i = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        self._h = self._lib.shim_open(find_library().encode(), v.ctypes.data, len(v), i.ctypes.data, len(i))
        if not self._h:
            raise RuntimeError("embree_adapter: cannot bind %s (not Embree 3/4?)" % find_library())
        self.version = self._lib.shim_version(self._h)

    def trace(self, rays, threads=1):
        r = np.ascontiguousarray(rays).view(RAY_DTYPE).reshape(-1)
        out = np.zeros(len(r), RESULT_DTYPE)
        self._lib.shim_trace(self._h, r.ctypes.data, out.ctypes.data, len(r), threads)
        return out

    def close(self):
        if self._h:
            self._lib.shim_close(self._h)
            self._h = None


def trace(sc, rays, threads=1):
    s = Scene(sc["vertices"], sc["indices"])
    try:
        return s.trace(rays, threads)
    finally:
        s.close()


def time_batch(sc, rays, threads):
    s = Scene(sc["vertices"], sc["indices"])
    try:
        s.trace(rays, threads)
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            s.trace(rays, threads)
            times.append(time.perf_counter() - t0)
        return {"value": round(len(rays) / float(np.median(times)) / 1e6, 2), "unit": "Mrays/s", "cores": threads, "kind": "reference-dependency",
                "what": "system Embree %d.x (%s) through oracle/embree_shim.c: rtcIntersect1 per ray, 1024-ray slices over %d pthreads "
                        "(the reference: Embree 2.x rtcIntersect8, Scene.cpp:386-428)" % (s.version, find_library(), threads)}
    finally:
        s.close()
