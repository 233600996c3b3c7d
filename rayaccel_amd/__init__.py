"""rayaccel_amd — MI355X (gfx950) wavefront ray-intersection engine.

Drop-in for the intersect-batch hot path of rasmusbarr/rayaccel's RayAccelerator
(Ray stream in, Result out).  The product is `libracc_hip.so` (hand-written HIP +
a C-ABI, include/racc_hip.h); this package is the host-side mirror used by the
tests and the bench harness.
"""
from . import synth  # noqa: F401
from .engine import (Comm, Context, Group, DeviceBuffer, LANE_AUTO, Environment, HostScene, RaccError, Scene,  # noqa: F401
                     RAY_DTYPE, RESULT_DTYPE, INVALID_TRIANGLE, build_library, device_count, load_library)
