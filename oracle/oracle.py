"""ctypes loader for the CPU restatement (oracle/racc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by anything under rayaccel_amd/.
Pinned against the reference's own traversal kernel (oracle/_ref, oracle/ref_kernel.py); see racc_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libracc_oracle.so")

RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("minT", "<f4"), ("dir", "<f4", 3), ("maxT", "<f4")], align=False)
RESULT_DTYPE = np.dtype([("triangle", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")], align=False)
BVH2_NODE_DTYPE = np.dtype([("kind", "<u4"), ("parent", "<u4"), ("first", "<u4"), ("last", "<u4"),
                            ("bbMin", "<f4", 3), ("dummy0", "<u4"), ("bbMax", "<f4", 3), ("dummy1", "<u4")])
GPU_NODE_DTYPE = np.dtype([("kind", "<u4"), ("parent", "<u4"), ("first", "<u4"), ("last", "<u4"),
                           ("leftMin", "<f4", 3), ("leftMax", "<f4", 3), ("rightMin", "<f4", 3), ("rightMax", "<f4", 3)])
PAIR_DTYPE = np.dtype([("e1", "<f4", 3), ("e3x", "<f4"), ("e2", "<f4", 3), ("e3y", "<f4"), ("p0", "<f4", 3), ("e3z", "<f4")])
assert RAY_DTYPE.itemsize == 32 and RESULT_DTYPE.itemsize == 16
assert BVH2_NODE_DTYPE.itemsize == 48 and GPU_NODE_DTYPE.itemsize == 64 and PAIR_DTYPE.itemsize == 48


def build(force=False):
    """Compile oracle/libracc_oracle.so with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "racc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "racc_oracle_simd.c")), os.path.getmtime(os.path.join(_HERE, "racc_oracle_simd512.c"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libracc_oracle.so"])
    # oracle/_ref: the reference's own traversal kernel, built only where /root/reference exists (this container)
    subprocess.call(["make", "-s", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        vp, u32 = C.c_void_p, C.c_uint32
        _lib.orc_bvh2_build.argtypes = [vp, u32, vp, u32, vp, vp, vp]
        _lib.orc_bvh2_build.restype = C.c_int
        _lib.orc_scene_pack.argtypes = [vp, u32, vp, vp, vp, u32, vp, vp, vp, vp, vp, vp]
        _lib.orc_scene_pack.restype = C.c_int
        _lib.orc_traverse.argtypes = [vp, vp, vp, vp, u32, u32, vp, vp, u32, u32, vp, vp, vp]
        _lib.orc_traverse.restype = None
        _lib.orc_traverse_mt.argtypes = [vp, vp, vp, vp, u32, u32, vp, vp, u32, u32, u32, u32]
        _lib.orc_traverse_mt.restype = None
        _lib.orc_traverse_simd_mt.argtypes = [vp, vp, vp, vp, u32, u32, vp, vp, u32, u32, u32, u32, u32]
        _lib.orc_simd512_available.restype = C.c_int
        _lib.orc_traverse_simd_mt.restype = None
        _lib.orc_simd_available.restype = C.c_int
        _lib.orc_env_sample.argtypes = [vp, u32, u32, vp, vp]
        _lib.orc_env_sample.restype = None
        _lib.orc_brute_closest.argtypes = [vp, vp, u32, vp, u32, vp, vp, vp, vp, vp]
        _lib.orc_brute_closest.restype = None
        _lib.orc_brute_one.argtypes = [vp, vp, u32, vp, vp, vp, vp]
        _lib.orc_brute_one.restype = C.c_int
        _lib.orc_pt_sample_material.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp]
        _lib.orc_pt_sample_material.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _verts4(vertices):
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    if v.ndim != 2 or v.shape[1] not in (3, 4):
        raise ValueError("vertices must be [V,3] or [V,4]")
    if v.shape[1] == 3:
        v = np.concatenate([v, np.zeros((len(v), 1), np.float32)], axis=1)
    return np.ascontiguousarray(v)


def bvh2_build(vertices, indices):
    """Bvh2.cpp restatement -> (nodes[nodeCount] BVH2_NODE_DTYPE, triangles[T] u32)."""
    v = _verts4(vertices)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    T = idx.size // 3
    nodes = np.zeros(2 * T, BVH2_NODE_DTYPE)
    tris = np.zeros(T, np.uint32)
    n = C.c_uint32(0)
    rc = lib().orc_bvh2_build(_p(v), len(v), _p(idx), T, _p(nodes), _p(tris), C.byref(n))
    if rc:
        raise RuntimeError("orc_bvh2_build failed: %d" % rc)
    return nodes[: n.value].copy(), tris


def scene_pack(nodes, triangles, vertices, indices):
    """Scene.cpp:237-339 restatement -> dict(nodes, pairs (padded), remap, pair_count)."""
    v = _verts4(vertices)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    T = idx.size // 3
    nodes = nodes.copy()
    gpu = np.zeros(len(nodes), GPU_NODE_DTYPE)
    pairs = np.zeros(T + 64, PAIR_DTYPE)
    remap = np.zeros(2 * T, np.uint32)
    ng, npair, npad = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    rc = lib().orc_scene_pack(_p(nodes), len(nodes), _p(triangles), _p(v), _p(idx), T,
                              _p(gpu), C.byref(ng), _p(pairs), C.byref(npair), C.byref(npad), _p(remap))
    if rc:
        raise RuntimeError("orc_scene_pack failed: %d" % rc)
    return dict(nodes=gpu[: ng.value].copy(), pairs=pairs[: npad.value].copy(),
                remap=remap[: 2 * npair.value].copy(), pair_count=npair.value)


def build_scene(vertices, indices):
    """createScene's GPU branch (Scene.cpp:216-349): build + pack."""
    nodes, tris = bvh2_build(vertices, indices)
    return scene_pack(nodes, tris, vertices, indices)


def traverse(scene, rays, env=None, counters=False, threads=1, repeat=1, out=None):
    """Kernels.h `traversal` restatement.  rays: RAY_DTYPE[N] -> RESULT_DTYPE[N]
    (+ nv, np, depth uint32[N] when counters)."""
    rays = np.ascontiguousarray(rays)
    assert rays.dtype == RAY_DTYPE
    n = len(rays)
    if out is None:
        out = np.zeros(n, RESULT_DTYPE)
    envp, w, h = None, 0, 0
    if env is not None:
        env = np.ascontiguousarray(env, dtype=np.float32)
        h, w = env.shape[0], env.shape[1]
        envp = _p(env)
    nodes, pairs, remap = scene["nodes"], scene["pairs"], scene["remap"]
    if counters:
        nv, npp, dp = (np.zeros(n, np.uint32) for _ in range(3))
        if threads > 1 and n >= 4096:
            # orc_traverse takes a ray range and ctypes releases the GIL: slices on Python threads (per-ray counters, so no sharing)
            import threading
            L = lib()
            cuts = [n * k // threads for k in range(threads + 1)]
            ts = [threading.Thread(target=L.orc_traverse, args=(_p(nodes), _p(pairs), _p(remap), envp, w, h, _p(rays), _p(out), cuts[k], cuts[k + 1], _p(nv), _p(npp), _p(dp)))
                  for k in range(threads)]
            [t.start() for t in ts]
            [t.join() for t in ts]
        else:
            lib().orc_traverse(_p(nodes), _p(pairs), _p(remap), envp, w, h, _p(rays), _p(out), 0, n, _p(nv), _p(npp), _p(dp))
        return out, nv, npp, dp
    if threads > 1 or repeat > 1:
        lib().orc_traverse_mt(_p(nodes), _p(pairs), _p(remap), envp, w, h, _p(rays), _p(out), n, 1024, threads, repeat)
    else:
        lib().orc_traverse(_p(nodes), _p(pairs), _p(remap), envp, w, h, _p(rays), _p(out), 0, n, None, None, None)
    return out


def simd_available():
    """The host has AVX2 + FMA (racc_oracle_simd.c needs them)."""
    return bool(lib().orc_simd_available())


def simd_width():
    """Widest form the host supports: 16 (AVX-512F/DQ/VL, racc_oracle_simd512.c), 8 (AVX2 + FMA, racc_oracle_simd.c) or 0."""
    return 16 if lib().orc_simd512_available() else (8 if simd_available() else 0)


def traverse_simd(scene, rays, env=None, threads=1, repeat=1, out=None, width=8):
    """orc_traverse eight (AVX2) or sixteen (AVX-512) rays at a time: bit-identical to traverse(); bench.py's cpu_baseline kind "simd-port"."""
    rays = np.ascontiguousarray(rays)
    assert rays.dtype == RAY_DTYPE
    n = len(rays)
    if out is None:
        out = np.zeros(n, RESULT_DTYPE)
    envp, w, h = None, 0, 0
    if env is not None:
        env = np.ascontiguousarray(env, dtype=np.float32)
        h, w = env.shape[0], env.shape[1]
        envp = _p(env)
    lib().orc_traverse_simd_mt(_p(scene["nodes"]), _p(scene["pairs"]), _p(scene["remap"]), envp, w, h, _p(rays), _p(out), n, 1024, threads, repeat, width)
    return out


def env_sample(env, dirs):
    env = np.ascontiguousarray(env, dtype=np.float32)
    dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((len(dirs), 3), np.float32)
    for i in range(len(dirs)):
        lib().orc_env_sample(_p(env), env.shape[1], env.shape[0], _p(dirs[i]), _p(out[i]))
    return out


def brute_closest(vertices, indices, rays):
    v = _verts4(vertices)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    rays = np.ascontiguousarray(rays)
    n = len(rays)
    tri = np.zeros(n, np.uint32)
    t, u, vv, t2 = (np.zeros(n, np.float64) for _ in range(4))
    lib().orc_brute_closest(_p(v), _p(idx), idx.size // 3, _p(rays), n, _p(tri), _p(t), _p(u), _p(vv), _p(t2))
    return tri, t, u, vv, t2


def brute_one(vertices, indices, triangle, ray):
    v = _verts4(vertices)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    ray = np.ascontiguousarray(ray).reshape(1)
    t, u, vv = C.c_double(0), C.c_double(0), C.c_double(0)
    hit = lib().orc_brute_one(_p(v), _p(idx), int(triangle), _p(ray), C.byref(t), C.byref(u), C.byref(vv))
    return bool(hit), t.value, u.value, vv.value


def algorithmic_bytes(results, nv, npairs):
    """B(ray) = 32 + 16 + 64*Nv + 48*Np + 4*[hit]."""
    hit = (results["triangle"] != 0xFFFFFFFF).astype(np.int64)
    return 48 * len(results) + 64 * int(nv.astype(np.int64).sum()) + 48 * int(npairs.astype(np.int64).sum()) + 4 * int(hit.sum())


def pt_sample_material(ke, rnd, normal, wo, exact_trig=False):
    """One lane of ReflectiveDiffuseMaterial::sample8 (Renderer/Materials.cpp:39-151) per sample -> (wi[n,3], colour[n,3], diffuse[n])."""
    ke = np.ascontiguousarray(ke, np.float32)
    rnd, normal, wo = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (rnd, normal, wo))
    n = len(rnd)
    wi, colour, diffuse = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
    for i in range(n):
        diffuse[i] = lib().orc_pt_sample_material(_p(ke), _p(rnd[i]), _p(normal[i]), _p(wo[i]), 1 if exact_trig else 0, _p(wi[i]), _p(colour[i]))
    return wi, colour, diffuse
