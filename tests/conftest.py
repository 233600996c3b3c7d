import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle and the HIP library exist (hipcc cross-compiles on CPU-only hosts)."""
    from oracle import oracle
    import rayaccel_amd

    oracle.build()
    rayaccel_amd.build_library()
    yield


@pytest.fixture(scope="session")
def small_scene():
    from rayaccel_amd import synth
    return synth.battlefield_synth(grid=40, boxes=32, quads=100)


@pytest.fixture(scope="session")
def small_host(small_scene):
    import rayaccel_amd
    return rayaccel_amd.HostScene(small_scene["vertices"], small_scene["indices"])


@pytest.fixture(scope="session")
def small_default_host(small_scene):
    """The small scene as racc::createScene builds it (library default options: the quality-1 tree, RACC_HOST_BUILD_DEFAULT_QUALITY)."""
    import rayaccel_amd
    return rayaccel_amd.HostScene(small_scene["vertices"], small_scene["indices"], quality=None)


@pytest.fixture(scope="session")
def full_default_blobs(full):
    """battlefield-synth as racc::createScene builds it (library default options), for tests that re-trace what racc::render traced."""
    import rayaccel_amd as ra
    return ra.HostScene(full["sc"]["vertices"], full["sc"]["indices"], quality=None).blobs()


@pytest.fixture(scope="session")
def gpu_ctx():
    import rayaccel_amd
    # chain_min_rays=1: the suite's shared context chains EVERY device-resident batch, as rounds 2-3 did, so that the chained path keeps
    # being exercised with batches of every size (the default chains only batches of >= 786,432 rays: racc_hip_options::chain_min_rays;
    # tests/test_gpu_parity.py::test_chained_launches also runs the default)
    ctx = rayaccel_amd.Context(device=0, chain_min_rays=1)     # raises (never falls back) when the extension or GPU is missing
    yield ctx
    ctx.destroy()


@pytest.fixture(scope="session")
def full(gpu_ctx):
    """battlefield-synth at full size (1.07 M triangles) on the GPU, with the 1M-ray coherent primary batch."""
    import rayaccel_amd as ra
    from rayaccel_amd import synth
    sc = synth.battlefield_synth()
    host = ra.HostScene(sc["vertices"], sc["indices"])          # quality 0: the reference builder's tree (leaves of up to six pairs)
    scene = gpu_ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = gpu_ctx.create_environment(sc["env"])
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    yield dict(sc=sc, host=host, blobs=host.blobs(), scene=scene, env=env, primary=prim)
    scene.destroy(); env.destroy()


@pytest.fixture(scope="session")
def small(gpu_ctx, small_scene, small_host):
    """The small test scene on the GPU with its 256x256 primary batch."""
    from rayaccel_amd import synth
    scene = gpu_ctx.upload_scene(small_host.nodes, small_host.pairs, small_host.remap)
    env = gpu_ctx.create_environment(small_scene["env"])
    prim, _ = synth.primary_rays(small_scene["camera"], 256, 256)
    yield dict(scene=scene, env=env, blobs=small_host.blobs(), primary=prim, sc=small_scene)
    scene.destroy()
    env.destroy()
