"""ctypes binding of include/racc_hip.h, shaped like the reference's `racc::` interface.

Python is only the test/bench harness language here; the product is
rayaccel_amd/libracc_hip.so (hand-written HIP for gfx950 behind a C-ABI).  The
names below mirror RayAccelerator/RayAccelerator.h:95-115 for the intersect-batch
path: `Context` ≙ racc::createContext, `Context.create_scene` ≙ racc::createScene,
`Context.create_environment` ≙ racc::createEnvironment, `Context.intersect` ≙ one
gpuWorkerThread dispatch (RayAccelerator.cpp:378-404).

There is NO CPU fallback: if the library is missing, fails to load, or finds no
gfx950 device, every GPU entry point raises RaccError.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .synth import RAY_DTYPE, RESULT_DTYPE, INVALID_TRIANGLE  # noqa: F401  (re-exported)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libracc_hip.so")
API_LIB_PATH = os.path.join(_HERE, "librayaccelerator.so")       # racc:: C++ interface over the C-ABI
PT_LIB_PATH = os.path.join(_HERE, "libracc_pathtracer.so")       # path-tracing consumer (BASELINE configs[4]), host shading
PTDEV_LIB_PATH = os.path.join(_HERE, "libracc_ptdev.so")         # the same consumer with generation/shading kernels on the GPU
CSRC = os.path.join(_HERE, "csrc")

BVH2_NODE_DTYPE = np.dtype([("kind", "<u4"), ("parent", "<u4"), ("first", "<u4"), ("last", "<u4"),
                            ("bbMin", "<f4", 3), ("dummy0", "<u4"), ("bbMax", "<f4", 3), ("dummy1", "<u4")])
GPU_NODE_DTYPE = np.dtype([("kind", "<u4"), ("parent", "<u4"), ("first", "<u4"), ("last", "<u4"),
                           ("leftMin", "<f4", 3), ("leftMax", "<f4", 3), ("rightMin", "<f4", 3), ("rightMax", "<f4", 3)])
PAIR_DTYPE = np.dtype([("e1", "<f4", 3), ("e3x", "<f4"), ("e2", "<f4", 3), ("e3y", "<f4"), ("p0", "<f4", 3), ("e3z", "<f4")])


class RaccError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("racc_hip error %d: %s" % (code, message))
        self.code = code


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("lanes", C.c_uint32), ("waves_per_simd", C.c_uint32),
                ("kernel_variant", C.c_uint32), ("refill_min", C.c_uint32), ("leaf_min", C.c_uint32),
                ("chunk", C.c_uint32), ("tail_active", C.c_uint32), ("regroup_period", C.c_uint32), ("thin_reps", C.c_uint32), ("inner_reps", C.c_uint32), ("coop_same_pct", C.c_uint32), ("time_kernels", C.c_uint32), ("drain_prefetch", C.c_uint32), ("leaf_step", C.c_uint32), ("wide_below", C.c_uint32), ("chain_launches", C.c_uint32), ("chain_min_rays", C.c_uint32)]


class SceneInfo(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("pair_count", C.c_uint32), ("remap_count", C.c_uint32),
                ("inner_height", C.c_uint32), ("max_leaf_pairs", C.c_uint32), ("spill_levels", C.c_uint32),
                ("device_bytes", C.c_uint64)]


class LaunchInfo(C.Structure):
    _fields_ = [("grid_blocks", C.c_uint32), ("block_threads", C.c_uint32), ("lds_bytes_per_block", C.c_uint32),
                ("waves_per_simd", C.c_uint32), ("last_kernel_ms", C.c_float)]


# Every symbol include/racc_hip.h declares, with its ctypes signature.
_vp, _u32, _u64, _i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
_P = C.POINTER
ABI = {
    "racc_hip_last_error": (C.c_char_p, []),
    "racc_hip_version": (C.c_char_p, []),
    "racc_hip_variant_available": (_i, [_u32]),
    "racc_hip_lane_count": (_i, [_vp, _P(_u32), _P(_u32)]),
    "racc_hip_device_count": (_i, [_P(_i)]),
    "racc_hip_create": (_i, [_i, _P(Options), _P(_vp)]),
    "racc_hip_destroy": (_i, [_vp]),
    "racc_hip_scene_upload": (_i, [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _P(_vp)]),
    "racc_hip_scene_free": (_i, [_vp, _vp]),
    "racc_hip_scene_get_info": (_i, [_vp, _P(SceneInfo)]),
    "racc_hip_env_upload": (_i, [_vp, _vp, _u32, _u32, _P(_vp)]),
    "racc_hip_env_free": (_i, [_vp, _vp]),
    "racc_hip_register_stream": (_i, [_vp, _vp, _vp, _u32]),
    "racc_hip_unregister_stream": (_i, [_vp, _vp, _vp]),
    "racc_hip_register_host": (_i, [_vp, _vp, _u64]),
    "racc_hip_unregister_host": (_i, [_vp, _vp]),
    "racc_hip_intersect": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _u32]),
    "racc_hip_intersect_async": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _u32]),
    "racc_hip_wait": (_i, [_vp, _u32]),
    "racc_hip_intersect_streams": (_i, [_vp, _vp, _vp, _u32, _P(_vp), _P(_vp), _P(_u32), _u32]),
    "racc_hip_intersect_streams_async": (_i, [_vp, _vp, _vp, _u32, _P(_vp), _P(_vp), _P(_u32), _u32]),
    "racc_hip_intersect_device": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    "racc_hip_intersect_device_timed": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _P(C.c_float)]),
    "racc_hip_get_launch_info": (_i, [_vp, _u32, _P(LaunchInfo)]),
    "racc_hip_read_kernel_times": (_i, [_vp, _u32, _P(C.c_float), _u32, _P(_u32)]),
    "racc_hip_read_stats": (_i, [_vp, _u32, _P(_u64), _i]),
    "racc_hip_malloc": (_i, [_vp, _u64, _P(_vp)]),
    "racc_hip_free": (_i, [_vp, _vp]),
    "racc_hip_memcpy_h2d": (_i, [_vp, _vp, _vp, _u64]),
    "racc_hip_memcpy_d2h": (_i, [_vp, _vp, _vp, _u64]),
    "racc_hip_memcpy_d2d_async": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "racc_hip_synchronize": (_i, [_vp]),
    "racc_hip_stream_create": (_i, [_vp, _P(_vp)]),
    "racc_hip_stream_synchronize": (_i, [_vp, _vp]),
    "racc_hip_stream_destroy": (_i, [_vp, _vp]),
    "racc_hip_group_create": (_i, [_P(_i), _u32, _P(Options), _P(_vp)]),
    "racc_hip_group_destroy": (_i, [_vp]),
    "racc_hip_group_size": (_u32, [_vp]),
    "racc_hip_group_ctx": (_vp, [_vp, _u32]),
    "racc_hip_group_scene_upload": (_i, [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _P(_vp)]),
    "racc_hip_group_scene_free": (_i, [_vp, _vp]),
    "racc_hip_group_env_upload": (_i, [_vp, _vp, _u32, _u32, _P(_vp)]),
    "racc_hip_group_env_free": (_i, [_vp, _vp]),
    "racc_hip_group_intersect": (_i, [_vp, _vp, _vp, _vp, _vp, _u32]),
    "racc_hip_group_intersect_device": (_i, [_vp, _vp, _vp, _P(_vp), _P(_vp), _P(_u32)]),
    "racc_hip_group_wait": (_i, [_vp]),
    "racc_hip_comm_unique_id": (_i, [_vp]),
    "racc_hip_comm_init_rank": (_i, [_vp, _vp, _i, _i, _P(_vp)]),
    "racc_hip_allgather_results": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "racc_hip_comm_destroy": (_i, [_vp]),
    "racc_host_scene_build": (_i, [_vp, _u32, _vp, _u32, _P(_vp)]),
    "racc_host_scene_build_ex": (_i, [_vp, _u32, _vp, _u32, _vp, _P(_vp)]),
    "racc_host_scene_free": (_i, [_vp]),
    "racc_host_scene_blobs": (_i, [_vp, _P(_vp), _P(_u32), _P(_vp), _P(_u32), _P(_u32), _P(_vp), _P(_u32)]),
    "racc_host_scene_bvh2": (_i, [_vp, _P(_vp), _P(_u32), _P(_vp), _P(_u32)]),
    "racc_host_scene_device_nodes": (_i, [_vp, _u32, _u32, _u32, _i, _vp, _u32, _P(_u32)]),
}

_lib = None


def build_library(force=False):
    """Compile libracc_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    outs = [LIB_PATH, API_LIB_PATH, PT_LIB_PATH, PTDEV_LIB_PATH, os.path.join(_HERE, "..", "tests", "cpp", "render_check")]
    if force:
        for o in outs:
            if os.path.exists(o):
                os.remove(o)
    subprocess.check_call(["make", "-s", "-C", CSRC])      # make knows every dependency (the kernels' .inc files included): no second list here
    return LIB_PATH


def available_variants(upto=64):
    """kernel_variant numbers this build of the library accepts (the shipped build: the V8 rows only)."""
    lib = load_library()
    return [v for v in range(1, upto + 1) if lib.racc_hip_variant_available(v)]


def load_library():
    """dlopen libracc_hip.so and bind every ABI symbol.  Raises RaccError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RaccError(-3, "libracc_hip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RaccError(-3, "cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)  # AttributeError here means the header and the .so disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise RaccError(rc, load_library().racc_hip_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _as_verts4(vertices):
    v = np.asarray(vertices, dtype=np.float32)
    if v.ndim != 2 or v.shape[1] not in (3, 4):
        raise ValueError("vertices must be [V,3] or [V,4] float32")
    if v.shape[1] == 3:
        v = np.concatenate([v, np.zeros((len(v), 1), np.float32)], 1)
    out = np.empty((len(v) + 4, 4), np.float32)  # over-allocate to find a 16-byte aligned start
    off = (-out.ctypes.data % 16) // 4
    flat = out.reshape(-1)[off: off + v.size].reshape(v.shape)
    flat[...] = v
    return flat


def device_count():
    n = C.c_int(0)
    rc = load_library().racc_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


LIBRARY_DEFAULT_QUALITY = 1      # RACC_HOST_BUILD_DEFAULT_QUALITY (include/racc_hip.h); tests/test_abi.py holds the two together


class HostBuildOptions(C.Structure):
    """racc_host_build_options (include/racc_hip.h)."""
    _fields_ = [("struct_size", _u32), ("quality", _u32), ("threads", _u32), ("split_percent", _u32), ("reserved", _u32 * 4)]


class HostScene:
    """Host-side build product ≙ the GPU branch of racc::createScene (Scene.cpp:216-339).  quality 0 (this harness's default: the tree the
    oracle's builder restates byte for byte) = the reference's builder; 1 / 2 = the same format with fewer node visits per ray; an integer
    is always passed as explicit options (racc_host_scene_build_ex), so no environment variable can change what it means.  quality=None =
    the LIBRARY's default — what racc::createScene and every other caller of the plain racc_host_scene_build gets: quality 1 since round 6
    (RACC_HOST_BUILD_DEFAULT_QUALITY), RACC_BUILD_QUALITY overrides it; `self.quality` then says which tree was built."""

    def __init__(self, vertices, indices, quality=0, threads=0, split_percent=0):
        """split_percent (quality >= 1): the spatial-split budget, extra triangle references as a percentage of the triangle count;
        0 = the library default (RACC_HOST_BUILD_DEFAULT_SPLIT_PERCENT = 10; RACC_BUILD_SPLIT_PERCENT overrides), -1 = no splits."""
        lib = load_library()
        v = _as_verts4(vertices)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        h = C.c_void_p()
        if quality is None and not threads and not split_percent:
            env = os.environ.get("RACC_BUILD_QUALITY")
            self.quality = max(0, int(env)) if env not in (None, "") else LIBRARY_DEFAULT_QUALITY
            _check(lib.racc_host_scene_build(_ptr(v), len(v), _ptr(idx), idx.size, C.byref(h)))
        else:
            if quality is None:
                env = os.environ.get("RACC_BUILD_QUALITY")
                quality = max(0, int(env)) if env not in (None, "") else LIBRARY_DEFAULT_QUALITY
            self.quality = int(quality)
            opt = HostBuildOptions(struct_size=C.sizeof(HostBuildOptions), quality=int(quality), threads=int(threads),
                                   split_percent=(0xFFFFFFFF if int(split_percent) < 0 else int(split_percent)))
            _check(lib.racc_host_scene_build_ex(_ptr(v), len(v), _ptr(idx), idx.size, C.byref(opt), C.byref(h)))
        try:
            pn, pp, pr = C.c_void_p(), C.c_void_p(), C.c_void_p()
            nn, npad, npair, nr = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
            _check(lib.racc_host_scene_blobs(h, C.byref(pn), C.byref(nn), C.byref(pp), C.byref(npad), C.byref(npair), C.byref(pr), C.byref(nr)))
            self.nodes = np.ctypeslib.as_array(C.cast(pn, _P(C.c_uint8)), (nn.value * 64,)).view(GPU_NODE_DTYPE).copy()
            self.pairs = np.ctypeslib.as_array(C.cast(pp, _P(C.c_uint8)), (npad.value * 48,)).view(PAIR_DTYPE).copy()
            self.remap = np.ctypeslib.as_array(C.cast(pr, _P(C.c_uint32)), (nr.value,)).copy() if nr.value else np.zeros(0, np.uint32)
            self.pair_count = npair.value
            pb, pt = C.c_void_p(), C.c_void_p()
            nb, nt = C.c_uint32(), C.c_uint32()
            _check(lib.racc_host_scene_bvh2(h, C.byref(pb), C.byref(nb), C.byref(pt), C.byref(nt)))
            self.bvh_nodes = np.ctypeslib.as_array(C.cast(pb, _P(C.c_uint8)), (nb.value * 48,)).view(BVH2_NODE_DTYPE).copy()
            self.bvh_triangles = np.ctypeslib.as_array(C.cast(pt, _P(C.c_uint32)), (nt.value,)).copy()
        finally:
            lib.racc_host_scene_free(h)

    def blobs(self):
        return dict(nodes=self.nodes, pairs=self.pairs, remap=self.remap, pair_count=self.pair_count)

    def device_nodes(self, order=1):
        """The 64 B device records racc_hip_scene_upload lays the node blob out as: see device_nodes() below."""
        return device_nodes(self.nodes, len(self.pairs), len(self.remap), order)


def device_nodes(nodes, pair_count, remap_count, order=1):
    """The 64 B device records racc_hip_scene_upload lays a reference-format node blob out as (racc_host_scene_device_nodes; no GPU
    needed): [N', 16] uint32 words — child refs in words 0-1, the boxes as (min, max) plane pairs in words 4-15."""
    lib = load_library()
    nodes = np.ascontiguousarray(nodes)
    n = C.c_uint32(0)
    args = (_ptr(nodes), len(nodes), pair_count, remap_count, order)
    _check(lib.racc_host_scene_device_nodes(*args, None, 0, C.byref(n)))
    out = np.zeros((n.value, 16), np.uint32)
    _check(lib.racc_host_scene_device_nodes(*args, _ptr(out), n.value, C.byref(n)))
    return out


class Scene:
    def __init__(self, ctx, handle):
        self._ctx, self._h = ctx, handle
        info = SceneInfo()
        _check(load_library().racc_hip_scene_get_info(handle, C.byref(info)))
        self.info = {f: getattr(info, f) for f, _ in SceneInfo._fields_}

    def destroy(self):
        if self._h:
            load_library().racc_hip_scene_free(self._ctx._h, self._h)
            self._h = None


class Environment:
    def __init__(self, ctx, handle, width, height):
        self._ctx, self._h, self.width, self.height = ctx, handle, width, height

    def destroy(self):
        if self._h:
            load_library().racc_hip_env_free(self._ctx._h, self._h)
            self._h = None


class DeviceBuffer:
    """Device allocation owned by the engine (for hosts without torch)."""

    def __init__(self, ctx, nbytes):
        self._ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        _check(load_library().racc_hip_malloc(ctx._h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, array):
        a = np.ascontiguousarray(array)
        assert a.nbytes <= self.nbytes
        _check(load_library().racc_hip_memcpy_h2d(self._ctx._h, self.ptr, _ptr(a), a.nbytes))

    def download(self, dtype, count):
        out = np.empty(count, dtype)
        assert out.nbytes <= self.nbytes
        _check(load_library().racc_hip_memcpy_d2h(self._ctx._h, _ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            load_library().racc_hip_free(self._ctx._h, self.ptr)
            self.ptr = None


LANE_AUTO = 0xFFFFFFFF      # RACC_HIP_LANE_AUTO


class Context:
    """≙ racc::Context for the GPU intersect path; one per (process, GPU)."""

    def __init__(self, device=0, lanes=0, waves_per_simd=0, refill_min=0, leaf_min=0, chunk=0, kernel_variant=0, tail_active=0, regroup_period=0, thin_reps=0, inner_reps=0, coop_same_pct=0, time_kernels=0, drain_prefetch=0, leaf_step=0, wide_below=0, chain_launches=0, chain_min_rays=0):
        lib = load_library()
        o = Options()
        o.struct_size = C.sizeof(Options)
        o.lanes, o.waves_per_simd, o.refill_min, o.leaf_min, o.chunk, o.kernel_variant = lanes, waves_per_simd, refill_min, leaf_min, chunk, kernel_variant
        o.tail_active, o.regroup_period, o.thin_reps = tail_active, regroup_period, thin_reps
        o.inner_reps = inner_reps
        o.coop_same_pct = coop_same_pct
        o.time_kernels = time_kernels
        o.drain_prefetch = drain_prefetch
        o.leaf_step = leaf_step
        o.wide_below = wide_below
        o.chain_launches = chain_launches
        o.chain_min_rays = chain_min_rays
        h = C.c_void_p()
        _check(lib.racc_hip_create(device, C.byref(o), C.byref(h)))
        self._h = h
        n, rot = C.c_uint32(), C.c_uint32()
        _check(lib.racc_hip_lane_count(h, C.byref(n), C.byref(rot)))
        self.lanes, self.auto_lanes = int(n.value), int(rot.value)      # lanes the context has / lanes LANE_AUTO rotates over
        self.device = device

    _owned = True      # False: a group member's context, borrowed (Group.member): the group destroys it

    def destroy(self):
        if self._h and self._owned:
            load_library().racc_hip_destroy(self._h)
        self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    # -- scene / environment ---------------------------------------------------------------
    def upload_scene(self, nodes, pairs, remap):
        """Reference-format blobs in (≙ Scene.cpp:342-346)."""
        nodes = np.ascontiguousarray(nodes)
        pairs = np.ascontiguousarray(pairs)
        remap = np.ascontiguousarray(remap, dtype=np.uint32)
        assert nodes.dtype.itemsize == 64 and pairs.dtype.itemsize == 48
        h = C.c_void_p()
        _check(load_library().racc_hip_scene_upload(self._h, _ptr(nodes), len(nodes), _ptr(pairs), len(pairs), _ptr(remap), len(remap), C.byref(h)))
        return Scene(self, h)

    def create_scene(self, vertices, indices):
        """≙ racc::createScene (RayAccelerator.h:107): host build with the library's default options, then upload."""
        hs = HostScene(vertices, indices, quality=None)
        return self.upload_scene(hs.nodes, hs.pairs, hs.remap)

    def create_environment(self, colors):
        """≙ racc::createEnvironment (RayAccelerator.h:111); colors [H,W,4] float32."""
        c = np.ascontiguousarray(colors, dtype=np.float32)
        h = C.c_void_p()
        _check(load_library().racc_hip_env_upload(self._h, _ptr(c), c.shape[1], c.shape[0], C.byref(h)))
        return Environment(self, h, c.shape[1], c.shape[0])

    # -- intersect -------------------------------------------------------------------------
    def intersect(self, scene, env, rays, results=None, lane=0):
        """Host Ray[N] in, host Result[N] out, blocking (≙ enqueue + clFinish)."""
        rays = np.ascontiguousarray(rays)
        assert rays.dtype.itemsize == 32
        if results is None:
            results = np.zeros(len(rays), RESULT_DTYPE)
        _check(load_library().racc_hip_intersect(self._h, scene._h, env._h if env else None, _ptr(rays), _ptr(results), len(rays), lane))
        return results

    def intersect_async(self, scene, env, rays, results, lane=0):
        """Enqueue only (racc_hip_intersect_async): `rays` / `results` must stay alive and untouched until wait(lane)."""
        assert rays.dtype.itemsize == 32 and results.dtype.itemsize == 16 and rays.flags.c_contiguous and results.flags.c_contiguous
        _check(load_library().racc_hip_intersect_async(self._h, scene._h, env._h if env else None, _ptr(rays), _ptr(results), len(rays), lane))

    def register_host(self, array):
        """Page-locks a numpy array (racc_hip_register_host); returns a token for unregister_host."""
        _check(load_library().racc_hip_register_host(self._h, array.ctypes.data, array.nbytes))
        return array.ctypes.data

    def unregister_host(self, token):
        _check(load_library().racc_hip_unregister_host(self._h, token))

    def intersect_streams(self, scene, env, ray_arrays, lane=0):
        """Several host ray streams in ONE launch (≙ what racc::render hands the GPU thread)."""
        n = len(ray_arrays)
        rays = [np.ascontiguousarray(r) for r in ray_arrays]
        outs = [np.zeros(len(r), RESULT_DTYPE) for r in rays]
        pr = (C.c_void_p * n)(*[r.ctypes.data for r in rays])
        po = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        cn = (C.c_uint32 * n)(*[len(r) for r in rays])
        _check(load_library().racc_hip_intersect_streams(self._h, scene._h, env._h if env else None, n, pr, po, cn, lane))
        return outs

    def intersect_streams_async(self, scene, env, ray_arrays, out_arrays, lane=0):
        """Enqueue only (racc_hip_intersect_streams_async): the arrays must stay alive and untouched until wait(lane)."""
        n = len(ray_arrays)
        pr = (C.c_void_p * n)(*[r.ctypes.data for r in ray_arrays])
        po = (C.c_void_p * n)(*[o.ctypes.data for o in out_arrays])
        cn = (C.c_uint32 * n)(*[len(r) for r in ray_arrays])
        _check(load_library().racc_hip_intersect_streams_async(self._h, scene._h, env._h if env else None, n, pr, po, cn, lane))

    def intersect_device(self, scene, env, d_rays, d_results, count, lane=0, stream=None):
        _check(load_library().racc_hip_intersect_device(self._h, scene._h, env._h if env else None, d_rays, d_results, count, lane, stream))

    def intersect_device_timed(self, scene, env, d_rays, d_results, count, iters, lane=0):
        ms = (C.c_float * iters)()
        _check(load_library().racc_hip_intersect_device_timed(self._h, scene._h, env._h if env else None, d_rays, d_results, count, lane, iters, ms))
        return list(ms)

    def wait(self, lane=0):
        _check(load_library().racc_hip_wait(self._h, lane))

    def create_stream(self):
        st = C.c_void_p()
        _check(load_library().racc_hip_stream_create(self._h, C.byref(st)))
        return st.value

    def stream_synchronize(self, stream):
        _check(load_library().racc_hip_stream_synchronize(self._h, stream))

    def destroy_stream(self, stream):
        _check(load_library().racc_hip_stream_destroy(self._h, stream))

    def kernel_times(self, lane=0, capacity=256):
        """Durations (ms) of the lane's traversal kernels since the last call (Context(time_kernels=1))."""
        ms = (C.c_float * capacity)()
        n = C.c_uint32(0)
        _check(load_library().racc_hip_read_kernel_times(self._h, lane, ms, capacity, C.byref(n)))
        return list(ms[:n.value])

    def synchronize(self):
        _check(load_library().racc_hip_synchronize(self._h))

    def launch_info(self, lane=0):
        info = LaunchInfo()
        _check(load_library().racc_hip_get_launch_info(self._h, lane, C.byref(info)))
        return {f: getattr(info, f) for f, _ in LaunchInfo._fields_}

    def read_stats(self, lane=0, reset=True):
        st = (C.c_uint64 * 16)()
        _check(load_library().racc_hip_read_stats(self._h, lane, st, 1 if reset else 0))
        keys = ("inner_iters", "inner_lanes", "leaf_iters", "leaf_lanes", "refill_iters", "rays_loaded", "dequeues", "waves",
                "cy_inner", "cy_inner_load", "cy_leaf", "cy_leaf_load", "cy_refill", "cy_wave", "cy_shuffle", "shuffles")
        return dict(zip(keys, [int(x) for x in st]))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)


class Group:
    """The GPUs of one node behind one handle (racc_hip_group_*): scene replicated, host batches sharded, no exchange."""

    def __init__(self, devices):
        lib = load_library()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(lib.racc_hip_group_create(arr, len(devices), None, C.byref(h)))
        self._h, self.size, self.devices = h, lib.racc_hip_group_size(h), list(devices)
        self._scene = self._env = None

    def upload(self, nodes, pairs, remap, env_rgba=None):
        lib = load_library()
        nodes, pairs, remap = np.ascontiguousarray(nodes), np.ascontiguousarray(pairs), np.ascontiguousarray(remap, np.uint32)
        s = C.c_void_p()
        _check(lib.racc_hip_group_scene_upload(self._h, _ptr(nodes), len(nodes), _ptr(pairs), len(pairs), _ptr(remap), len(remap), C.byref(s)))
        self._scene = s
        if env_rgba is not None:
            img = np.ascontiguousarray(env_rgba, np.float32)
            e = C.c_void_p()
            _check(lib.racc_hip_group_env_upload(self._h, _ptr(img), img.shape[1], img.shape[0], C.byref(e)))
            self._env = e

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays)
        out = np.zeros(len(rays), RESULT_DTYPE)
        _check(load_library().racc_hip_group_intersect(self._h, self._scene, self._env, _ptr(rays), _ptr(out), len(rays)))
        return out

    def member(self, i):
        """The i-th member's engine context as a borrowed Context (device memory helpers, lanes)."""
        lib = load_library()
        c = Context.__new__(Context)
        c._h = C.c_void_p(lib.racc_hip_group_ctx(self._h, i))
        if not c._h:
            raise RaccError(-1, "group has no member %d" % i)
        c._owned = False          # destroy() / `with` on it only drops the reference: racc_hip_group_destroy frees the member
        n, rot = C.c_uint32(), C.c_uint32()
        _check(lib.racc_hip_lane_count(c._h, C.byref(n), C.byref(rot)))
        c.lanes, c.auto_lanes, c.device = int(n.value), int(rot.value), self.devices[i]
        return c

    def intersect_device(self, d_rays, d_results, counts):
        """Per-member device shards (lists of device pointers / ray counts); asynchronous, see wait()."""
        n = self.size
        pr = (C.c_void_p * n)(*d_rays); po = (C.c_void_p * n)(*d_results); cn = (C.c_uint32 * n)(*counts)
        _check(load_library().racc_hip_group_intersect_device(self._h, self._scene, self._env, pr, po, cn))

    def wait(self):
        _check(load_library().racc_hip_group_wait(self._h))

    def destroy(self):
        lib = load_library()
        if self._scene:
            lib.racc_hip_group_scene_free(self._h, self._scene)
        if self._env:
            lib.racc_hip_group_env_free(self._h, self._env)
        if self._h:
            lib.racc_hip_group_destroy(self._h)
        self._h = self._scene = self._env = None


class Comm:
    """RCCL communicator over the GPUs of one node, bound through the C-ABI (racc_hip_comm_*): all-gather of Result shards."""

    def __init__(self, ctx, unique_id, rank, nranks):
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(load_library().racc_hip_comm_init_rank(ctx._h, buf, rank, nranks, C.byref(h)))
        self._h, self.rank, self.nranks = h, rank, nranks

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(load_library().racc_hip_comm_unique_id(buf))
        return buf.raw

    def allgather_results(self, d_send, d_recv, count_per_rank, stream=None):
        _check(load_library().racc_hip_allgather_results(self._h, d_send, d_recv, count_per_rank, stream))

    def destroy(self):
        if self._h:
            load_library().racc_hip_comm_destroy(self._h)
            self._h = None


class PathTraceStats(C.Structure):
    _fields_ = [("rays_traced", C.c_uint64), ("primary_rays", C.c_uint64), ("seconds", C.c_double),
                ("tiles_x", C.c_uint32), ("tiles_y", C.c_uint32), ("max_depth", C.c_uint32), ("threads", C.c_uint32),
                ("triangles", C.c_uint32), ("reserved", C.c_uint32)]


def path_trace_capture_round(round_index, capacity):
    """Arms the device consumer's test hook (racc_ptdev_capture_round): the next path_trace(shading="gpu") copies the rays the
    engine was handed in bounce round `round_index` and the hits it returned.  Returns (rays, hits, count) — numpy arrays the
    render fills, `count` a ctypes uint32 holding how many records arrived."""
    lib = C.CDLL(PTDEV_LIB_PATH)
    rays, hits, count = np.zeros(capacity, RAY_DTYPE), np.zeros(capacity, RESULT_DTYPE), C.c_uint32(0)
    lib.racc_ptdev_capture_round.restype = None
    lib.racc_ptdev_capture_round.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.racc_ptdev_capture_round(round_index, _ptr(rays), _ptr(hits), capacity, C.byref(count))
    return rays, hits, count


def path_trace(scene_bin, width, height, spp_first, spp_count, device=0, max_depth=0, cpu_threads=0, shading="cpu", samples_per_batch=0):
    """Render samples [spp_first, spp_first+spp_count) of a reference-format scene file with the path-tracing consumer.
    shading="cpu": spawn/shade callbacks on host threads through racc::render (rayaccel_amd/csrc/pathtracer.cpp, the
    reference's shape); shading="gpu": generation, shading and compaction kernels around racc_hip_intersect_device, rays
    never leave HBM (rayaccel_amd/csrc/pt_device.hip).  Both render the same image bit for bit.
    Returns (sum of radiance [H,W,3] float64, stats dict)."""
    load_library()
    img = np.zeros((height, width, 3), np.float64)
    st = PathTraceStats()
    if shading == "cpu":
        lib = C.CDLL(PT_LIB_PATH)
        fn, last = lib.racc_pt_render_file, cpu_threads
    elif shading == "gpu":
        lib = C.CDLL(PTDEV_LIB_PATH)
        fn, last = lib.racc_ptdev_render_file, samples_per_batch
    else:
        raise ValueError("shading must be 'cpu' or 'gpu'")
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(PathTraceStats)]
    rc = fn(os.fsencode(scene_bin), device, width, height, spp_first, spp_count, max_depth, last, _ptr(img), C.byref(st))
    if rc != 0:
        raise RaccError(rc, "path tracer (%s shading) failed" % shading)
    return img, {f: getattr(st, f) for f, _ in PathTraceStats._fields_}
