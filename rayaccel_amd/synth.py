"""Deterministic synthetic inputs for the intersect-batch path.

`battlefield.bin`, the only scene every BASELINE.json config names, is absent
from the reference checkout (/root/reference/.MISSING_LARGE_BLOBS).  This module
writes a seeded stand-in, **battlefield-synth**, in the reference's own scene
file layout (Renderer/main.cpp:117-191) and generates the ray batches
SURVEY.md §8(d) defines: pinhole-coherent primaries in the reference's tile
order (Renderer/TiledRenderer.cpp:55-67, Renderer/Camera.cpp:13-26,55-85) and
first-bounce cosine-weighted diffuse rays (Renderer/PathTracingRenderer.cpp:
410-422, Renderer/Materials.cpp:96-121).  Everything is a pure function of
integer seeds through a counter-based hash, so the GPU box regenerates the
same bytes with no files shipped.

Every result produced from these inputs must be labelled "synthetic stand-in,
reference scene unavailable".
"""
import struct

import numpy as np

RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("minT", "<f4"), ("dir", "<f4", 3), ("maxT", "<f4")])
RESULT_DTYPE = np.dtype([("triangle", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
INVALID_TRIANGLE = 0xFFFFFFFF

SCENE_SEED = 0xBA77F1E1D & 0xFFFFFFFF
DIFFUSE_SEED = 0x5EED0001
TILE = 128  # Renderer/TiledRenderer.h:37


# ----------------------------------------------------------------- hashing
def pcg_hash(x):
    """PCG-RXS-M-XS 32-bit output hash, vectorised (uint32 in, uint32 out)."""
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    state = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return (((word >> 22) ^ word) & 0xFFFFFFFF).astype(np.uint32)


def hash_uniform(counter, stream, seed):
    """float32 in [0,1) from (counter, stream, seed)."""
    c = np.asarray(counter, dtype=np.uint64)
    h = pcg_hash((c + np.uint64(seed)) & 0xFFFFFFFF)
    h = pcg_hash((h.astype(np.uint64) ^ (np.uint64(stream) * np.uint64(0x9E3779B9))) & 0xFFFFFFFF)
    return ((h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


# ------------------------------------------------------------------- scene
def _height(x, z, seed):
    """Fractal height: 9 octaves of hashed-phase sinusoids (float64 -> float32)."""
    h = np.zeros_like(x, dtype=np.float64)
    amp, freq = 9.0, 0.035
    for k in range(9):
        th = float(hash_uniform(k, 11, seed)) * 2.0 * np.pi
        ph = float(hash_uniform(k, 13, seed)) * 2.0 * np.pi
        h += amp * np.sin(freq * (x * np.cos(th) + z * np.sin(th)) + ph)
        amp *= 0.55
        freq *= 1.9
    return h


def battlefield_synth(grid=700, boxes=4096, quads=20000, seed=SCENE_SEED, extent=100.0):
    """Stand-in scene: grid x grid-quad fractal height-field + axis-aligned boxes
    + thin random quads.  Defaults give 980,000 + 49,152 + 40,000 = 1,069,152
    triangles.  Returns dict(vertices [V,4] f32, indices [T,3]
    u32, camera, env [H,W,4] f32, name)."""
    n = grid + 1
    gx, gz = np.meshgrid(np.linspace(-extent, extent, n), np.linspace(-extent, extent, n), indexing="xy")
    gy = _height(gx, gz, seed)
    hv = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3)
    j, i = np.meshgrid(np.arange(grid), np.arange(grid), indexing="xy")
    a = (i * n + j).reshape(-1)
    b = a + 1
    c = a + n + 1
    d = a + n
    # (a,c,b) & (a,d,c): normals point +y; the two triangles share edge a-c reversed
    hidx = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)], axis=1).reshape(-1, 3)

    verts = [hv]
    idx = [hidx]
    base = len(hv)

    if boxes:
        k = np.arange(boxes)
        cx = (hash_uniform(k, 21, seed).astype(np.float64) * 2 - 1) * extent * 0.95
        cz = (hash_uniform(k, 22, seed).astype(np.float64) * 2 - 1) * extent * 0.95
        sx = 0.3 + 2.2 * hash_uniform(k, 23, seed).astype(np.float64)
        sy = 0.5 + 5.0 * hash_uniform(k, 24, seed).astype(np.float64) ** 2
        sz = 0.3 + 2.2 * hash_uniform(k, 25, seed).astype(np.float64)
        cy = _height(cx, cz, seed) + sy - 0.3
        corner = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1],
                           [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=np.float64)
        bv = (np.stack([cx, cy, cz], 1)[:, None, :] + corner[None] * np.stack([sx, sy, sz], 1)[:, None, :]).reshape(-1, 3)
        face = np.array([[0, 3, 2], [0, 2, 1], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4],
                         [3, 7, 6], [3, 6, 2], [0, 4, 7], [0, 7, 3], [1, 2, 6], [1, 6, 5]], dtype=np.int64)
        bi = (base + 8 * k[:, None, None] + face[None]).reshape(-1, 3)
        verts.append(bv)
        idx.append(bi)
        base += len(bv)

    if quads:
        k = np.arange(quads)
        cx = (hash_uniform(k, 31, seed).astype(np.float64) * 2 - 1) * extent * 0.9
        cz = (hash_uniform(k, 32, seed).astype(np.float64) * 2 - 1) * extent * 0.9
        cy = _height(cx, cz, seed) + 1.0 + 25.0 * hash_uniform(k, 33, seed).astype(np.float64)
        th = hash_uniform(k, 34, seed).astype(np.float64) * 2 * np.pi
        ph = np.arccos(1 - 2 * hash_uniform(k, 35, seed).astype(np.float64))
        nrm = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], 1)
        t0 = np.cross(nrm, np.where(np.abs(nrm[:, 1:2]) < 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]]))
        t0 /= np.linalg.norm(t0, axis=1, keepdims=True)
        t1 = np.cross(nrm, t0)
        lu = (0.2 + 3.0 * hash_uniform(k, 36, seed).astype(np.float64))[:, None]
        lv = (0.05 + 0.5 * hash_uniform(k, 37, seed).astype(np.float64))[:, None]
        ctr = np.stack([cx, cy, cz], 1)
        qv = np.stack([ctr - t0 * lu - t1 * lv, ctr + t0 * lu - t1 * lv,
                       ctr + t0 * lu + t1 * lv, ctr - t0 * lu + t1 * lv], 1).reshape(-1, 3)
        qf = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int64)
        qi = (base + 4 * k[:, None, None] + qf[None]).reshape(-1, 3)
        verts.append(qv)
        idx.append(qi)
        base += len(qv)

    v3 = np.concatenate(verts, 0).astype(np.float32)
    vertices = np.concatenate([v3, np.ones((len(v3), 1), np.float32)], 1)
    indices = np.concatenate(idx, 0).astype(np.uint32)

    eye = np.array([-0.92 * extent, float(gy.max()) + 0.28 * extent, -0.78 * extent], np.float32)
    camera = dict(origin=eye, target=np.array([0.12 * extent, float(gy.mean()) - 2.0, 0.1 * extent], np.float32),
                  up=np.array([0.0, 1.0, 0.0], np.float32), fov=55.0)
    return dict(vertices=np.ascontiguousarray(vertices), indices=np.ascontiguousarray(indices),
                camera=camera, env=environment_synth(), max_depth=5,
                name="battlefield-synth(grid=%d,boxes=%d,quads=%d,seed=0x%X)" % (grid, boxes, quads, seed))


XL_GRID = 3400


def battlefield_synth_xl(grid=XL_GRID, seed=SCENE_SEED):
    """battlefield-synth-XL (round 4): the same generator scaled past the MI355X's 256 MiB Infinity Cache — a 3400 x 3400-quad
    height-field, 96,632 boxes, 471,836 thin quads = 25,223,256 triangles, i.e. 594 MB of nodes + 623 MB of pairs + 104 MB of remap on
    the device (the reference format holds < 2^24 pairs, Scene.cpp:294-312: ~32 M triangles is its ceiling).  battlefield-synth
    (55 MB) lives in the L2s and the Infinity Cache after the first launch, so no roofline can bind there; this one is where the
    contractual HBM yard-stick is tested.  Same camera, same environment."""
    k = (grid / 700.0) ** 2
    sc = battlefield_synth(grid=grid, boxes=int(4096 * k) // 4 * 4, quads=int(20000 * k), seed=seed)
    sc["name"] = "battlefield-synth-XL(" + sc["name"].split("(", 1)[1]
    return sc


# ------------------------------------------------------- two more scene classes (round 6)
# battlefield-synth is one family: a connected height-field with small things on it.  These two have different statistics — what a tree
# builder and a traversal kernel tuned on one family must also survive (DESIGN.md §7 holds the three-row table).
_BOX_CORNERS = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=np.float64)
_BOX_FACES = np.array([[0, 3, 2], [0, 2, 1], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4],
                       [3, 7, 6], [3, 6, 2], [0, 4, 7], [0, 7, 3], [1, 2, 6], [1, 6, 5]], dtype=np.int64)


def city_synth(blocks=88, windows=(3, 5), seed=SCENE_SEED ^ 0xC17F, extent=100.0):
    """Axis-aligned "city": a ground plane of few LARGE triangles (8 x 8 quads over the whole extent), one box-shaped building per cell
    of a blocks x blocks street grid (heights spread over 1.5 decades), and on every facade a grid of SMALL window-sill quads standing
    2 cm proud of the wall.  Everything axis-aligned; triangle areas span seven orders of magnitude; nearly every ray that enters a
    street canyon grazes long coplanar walls.  Defaults: 128 + 7,744 x (12 + 4 x 15 x 2) = 1,022,336 triangles."""
    cell = 2.0 * extent / blocks
    g = np.linspace(-extent, extent, 9)
    gx, gz = np.meshgrid(g, g, indexing="xy")
    gv = np.stack([gx, np.zeros_like(gx), gz], -1).reshape(-1, 3)
    j, i = np.meshgrid(np.arange(8), np.arange(8), indexing="xy")
    a = (i * 9 + j).reshape(-1); b = a + 1; c = a + 10; d = a + 9
    verts, idx = [gv], [np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)], 1).reshape(-1, 3)]
    base = len(gv)
    k = np.arange(blocks * blocks)
    bx, bz = k % blocks, k // blocks
    cx = -extent + (bx + 0.5) * cell
    cz = -extent + (bz + 0.5) * cell
    hx = cell * (0.25 + 0.12 * hash_uniform(k, 61, seed).astype(np.float64))
    hz = cell * (0.25 + 0.12 * hash_uniform(k, 62, seed).astype(np.float64))
    hy = 0.5 * 10.0 ** (0.3 + 1.4 * hash_uniform(k, 63, seed).astype(np.float64) ** 2)          # half height: 1 .. 25
    ctr = np.stack([cx, hy, cz], 1)
    half = np.stack([hx, hy, hz], 1)
    bv = (ctr[:, None, :] + _BOX_CORNERS[None] * half[:, None, :]).reshape(-1, 3)
    verts.append(bv); idx.append((base + 8 * k[:, None, None] + _BOX_FACES[None]).reshape(-1, 3)); base += len(bv)
    wx, wy = windows
    if wx and wy:
        fu, fv = np.meshgrid((np.arange(wx) + 0.5) / wx * 2 - 1, (np.arange(wy) + 0.5) / wy * 2 - 1, indexing="xy")
        fu, fv = fu.reshape(-1), fv.reshape(-1)                                                 # window centres on a facade, in [-1, 1]^2
        su, sv = 0.55 / wx, 0.35 / wy
        quad = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=np.float64)
        for axis, sign in ((0, 1.0), (0, -1.0), (2, 1.0), (2, -1.0)):                           # facades +x, -x, +z, -z
            other = 2 - axis
            w = np.zeros((len(k), len(fu), 4, 3))
            w[..., axis] = (ctr[:, axis] + sign * (half[:, axis] + 0.02))[:, None, None]
            w[..., other] = ctr[:, other][:, None, None] + half[:, other][:, None, None] * (fu[None, :, None] + su * quad[None, None, :, 0])
            w[..., 1] = ctr[:, 1][:, None, None] + half[:, 1][:, None, None] * (fv[None, :, None] + sv * quad[None, None, :, 1])
            wv = w.reshape(-1, 3)
            q = np.arange(len(k) * len(fu))
            verts.append(wv); idx.append((base + 4 * q[:, None, None] + np.array([[0, 1, 2], [0, 2, 3]])[None]).reshape(-1, 3)); base += len(wv)
    v3 = np.concatenate(verts, 0).astype(np.float32)
    vertices = np.concatenate([v3, np.ones((len(v3), 1), np.float32)], 1)
    camera = dict(origin=np.array([-0.93 * extent, 0.42 * extent, -0.81 * extent], np.float32), target=np.array([0.1 * extent, 2.0, 0.05 * extent], np.float32),
                  up=np.array([0.0, 1.0, 0.0], np.float32), fov=55.0)
    return dict(vertices=np.ascontiguousarray(vertices), indices=np.ascontiguousarray(np.concatenate(idx, 0).astype(np.uint32)), camera=camera,
                env=environment_synth(), max_depth=5, name="city-synth(blocks=%d,windows=%dx%d,seed=0x%X)" % (blocks, wx, wy, seed))


def soup_synth(triangles=1000000, clusters=96, seed=SCENE_SEED ^ 0x50FA, extent=100.0):
    """Unconnected triangle soup with heavy overlap: no two triangles share a vertex (every pair of the packed scene is a lone triangle
    with the degenerate second one, Scene.cpp:174-178), sizes log-uniform over 0.05 .. 8, orientations uniform, centres in `clusters`
    blobs so that hundreds of triangles overlap in the middle of each — the sweep finds no cheap split there and closes LARGE leaves
    (Bvh2.cpp:462-485), which the kernel's multi-pair leaf loop must walk."""
    k = np.arange(triangles)
    cl = (pcg_hash(k ^ seed) % np.uint32(clusters)).astype(np.int64)
    cc = np.stack([(hash_uniform(np.arange(clusters), 71, seed) * 2 - 1) * extent * 0.85,
                   5.0 + hash_uniform(np.arange(clusters), 72, seed) * 40.0,
                   (hash_uniform(np.arange(clusters), 73, seed) * 2 - 1) * extent * 0.85], 1).astype(np.float64)
    rad = 2.0 + 14.0 * hash_uniform(np.arange(clusters), 74, seed).astype(np.float64)
    off = np.stack([hash_uniform(k, 75 + a, seed).astype(np.float64) + hash_uniform(k, 78 + a, seed).astype(np.float64) +
                    hash_uniform(k, 81 + a, seed).astype(np.float64) - 1.5 for a in range(3)], 1)      # ~ gaussian, sigma 0.5
    ctr = cc[cl] + off * rad[cl][:, None]
    size = 0.05 * 160.0 ** hash_uniform(k, 84, seed).astype(np.float64)
    # three random unit offsets around the centre, scaled: slivers included
    def unit(s0):
        z = hash_uniform(k, s0, seed).astype(np.float64) * 2 - 1
        ph = hash_uniform(k, s0 + 1, seed).astype(np.float64) * 2 * np.pi
        r = np.sqrt(np.maximum(0.0, 1 - z * z))
        return np.stack([r * np.cos(ph), z, r * np.sin(ph)], 1)
    tv = np.stack([ctr + unit(85) * size[:, None], ctr + unit(87) * size[:, None], ctr + unit(89) * size[:, None]], 1).reshape(-1, 3)
    v3 = tv.astype(np.float32)
    vertices = np.concatenate([v3, np.ones((len(v3), 1), np.float32)], 1)
    indices = np.arange(3 * triangles, dtype=np.uint32).reshape(-1, 3)
    camera = dict(origin=np.array([-1.25 * extent, 0.75 * extent, -1.1 * extent], np.float32), target=np.array([0.0, 18.0, 0.0], np.float32),
                  up=np.array([0.0, 1.0, 0.0], np.float32), fov=55.0)
    return dict(vertices=np.ascontiguousarray(vertices), indices=np.ascontiguousarray(indices), camera=camera, env=environment_synth(), max_depth=5,
                name="soup-synth(triangles=%d,clusters=%d,seed=0x%X)" % (triangles, clusters, seed))


SCENES = {"battlefield-synth": battlefield_synth, "city-synth": city_synth, "soup-synth": soup_synth}


def environment_synth(width=512, height=256):
    """512x256 RGBA32F gradient sky + sun lobe."""
    v, u = np.meshgrid((np.arange(height) + 0.5) / height, (np.arange(width) + 0.5) / width, indexing="ij")
    sky = np.stack([0.35 + 0.4 * v, 0.5 + 0.35 * v, 0.95 - 0.25 * v], -1)
    sun = np.exp(-((u - 0.7) ** 2 + (v - 0.3) ** 2) * 180.0)[..., None] * np.array([6.0, 5.2, 3.8])
    env = np.concatenate([sky + sun, np.ones((height, width, 1))], -1)
    return np.ascontiguousarray(env.astype(np.float32))


# ----------------------------------------------- reference .bin scene format
_HEADER = struct.Struct("<IIIHHHH3f3f3ff")  # Renderer/main.cpp:118-133, 60 bytes
assert _HEADER.size == 60


def write_scene_bin(path, scene, viewport=(1920, 1080)):
    """Write `scene` in the layout Renderer/main.cpp:117-191 reads."""
    v, idx, env, cam = scene["vertices"], scene["indices"], scene["env"], scene["camera"]
    T, V = len(idx), len(v)
    p = v[:, :3].astype(np.float64)
    fn = np.cross(p[idx[:, 1]] - p[idx[:, 0]], p[idx[:, 2]] - p[idx[:, 0]])
    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-30)
    vn = np.zeros((V, 3))
    for k in range(3):
        np.add.at(vn, idx[:, k], fn)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-30)
    with open(path, "wb") as f:
        f.write(_HEADER.pack(scene.get("max_depth", 5), V, T, viewport[0], viewport[1], env.shape[1], env.shape[0],
                             *map(float, cam["origin"]), *map(float, cam["target"]), *map(float, cam["up"]), float(cam["fov"])))
        f.write(idx.astype("<u4").tobytes())
        f.write((pcg_hash(np.arange(T)) & 3).astype("<u2").tobytes())
        f.write(np.concatenate([fn, np.zeros((T, 1))], 1).astype("<f4").tobytes())
        f.write(v.astype("<f4").tobytes())
        f.write(np.concatenate([vn, np.zeros((V, 1))], 1).astype("<f4").tobytes())
        f.write(np.zeros((V, 2), "<f4").tobytes())
        f.write(env.astype("<f4").tobytes())


def read_scene_bin(path):
    """Read a reference-format scene file (the real battlefield.bin drops in here)."""
    with open(path, "rb") as f:
        hdr = _HEADER.unpack(f.read(60))
        max_depth, V, T, vw, vh, ew, eh = hdr[:7]
        origin, target, up, fov = np.array(hdr[7:10], np.float32), np.array(hdr[10:13], np.float32), np.array(hdr[13:16], np.float32), hdr[16]
        idx = np.frombuffer(f.read(T * 12), "<u4").reshape(T, 3).copy()
        f.seek(T * 2 + T * 16, 1)
        v = np.frombuffer(f.read(V * 16), "<f4").reshape(V, 4).copy()
        f.seek(V * 16 + V * 8, 1)
        env = np.frombuffer(f.read(ew * eh * 16), "<f4").reshape(eh, ew, 4).copy()
    return dict(vertices=v, indices=idx, env=env, max_depth=max_depth, viewport=(vw, vh),
                camera=dict(origin=origin, target=target, up=up, fov=fov), name=path)


# -------------------------------------------------------------------- rays
def _normalize(v):
    return v / np.sqrt((v * v).sum(-1, keepdims=True))


def camera_basis(camera, width, height):
    """Camera::lookAt, Renderer/Camera.cpp:13-26 (float64 internally)."""
    o = np.asarray(camera["origin"], np.float64)
    fwd = _normalize(np.asarray(camera["target"], np.float64) - o)
    right = _normalize(np.cross(fwd, np.asarray(camera["up"], np.float64)))
    cup = np.cross(right, fwd)
    ey = np.tan(0.5 * camera["fov"] * np.pi / 180.0)
    ex = ey * (width / height)
    return o, right * (-2.0 / width * ex), cup * (-2.0 / height * ey), fwd + right * ex + cup * ey


def primary_rays(camera, width, height):
    """Pinhole-coherent batch: pixel-centre samples, exact-sqrt normalisation,
    minT=0, maxT=1e6, pixels ordered tile by tile (128x128, row-major inside a
    tile, tiles row-major) as TiledRenderer.cpp:55-67 + Camera.cpp:60-67.  Only
    floor(W/128) x floor(H/128) tiles exist (TiledRenderer.cpp:20-22)."""
    o, r, u, view = camera_basis(camera, width, height)
    tw, th = width // TILE, height // TILE
    ty, tx, y, x = np.meshgrid(np.arange(th), np.arange(tw), np.arange(TILE), np.arange(TILE), indexing="ij")
    px = (tx * TILE + x).reshape(-1)
    py = (ty * TILE + y).reshape(-1)
    d = view[None] + r[None] * (px[:, None] + 0.5) + u[None] * (py[:, None] + 0.5)
    d = _normalize(d)
    rays = np.zeros(len(px), RAY_DTYPE)
    rays["origin"] = o.astype(np.float32)
    rays["minT"] = 0.0
    rays["dir"] = d.astype(np.float32)
    rays["maxT"] = 1e6
    pixel = (py * width + px).astype(np.uint32)
    return rays, pixel


def diffuse_bounce_rays(scene, rays, results, count, seed=DIFFUSE_SEED, first_sample=0):
    """First-bounce diffuse batch from primary hits:
    o = P + 1e-4*Ng (Ng flipped toward the incoming side), d cosine-weighted
    about Ng from a counter-based hash of (ray index, sample), minT=1e-3,
    maxT=1e6.  Misses are skipped; the hit list is re-cycled with the next
    sample index until exactly `count` rays exist.  Order = primary order, i.e.
    spatially coherent origins with incoherent directions."""
    return diffuse_bounce_batches(scene, rays, results, count, [first_sample], seed)[0]


def diffuse_bounce_batches(scene, rays, results, count, first_samples, seed=DIFFUSE_SEED):
    """Several diffuse_bounce_rays batches (one per entry of `first_samples`) off the same primary hits: the hit points and
    tangent frames are computed once."""
    hit = np.nonzero(results["triangle"] != INVALID_TRIANGLE)[0]
    if len(hit) == 0:
        raise ValueError("no primary hits to bounce from")
    v = scene["vertices"][:, :3].astype(np.float64)
    tri = scene["indices"][results["triangle"][hit]]
    p0, p1, p2 = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
    ng = np.cross(p1 - p0, p2 - p0)
    ng /= np.maximum(np.linalg.norm(ng, axis=1, keepdims=True), 1e-30)
    d_in = rays["dir"][hit].astype(np.float64)
    ng = np.where(((ng * d_in).sum(1) > 0)[:, None], -ng, ng)
    P = rays["origin"][hit].astype(np.float64) + d_in * results["t"][hit].astype(np.float64)[:, None]
    helper = np.where(np.abs(ng[:, 0:1]) > 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
    bu = np.cross(helper, ng)
    bu /= np.linalg.norm(bu, axis=1, keepdims=True)
    bv = np.cross(ng, bu)
    origin32 = (P + 1e-4 * ng).astype(np.float32)

    batches = []
    for first_sample in first_samples:
        out = np.zeros(count, RAY_DTYPE)
        filled, sample = 0, first_sample
        while filled < count:
            n = min(len(hit), count - filled)
            ctr = hit[:n].astype(np.uint64) + np.uint64(sample) * np.uint64(0x01000193)
            r1 = hash_uniform(ctr, 41, seed)[:n].astype(np.float64) * 2 * np.pi
            r2 = hash_uniform(ctr, 42, seed)[:n].astype(np.float64)
            s = np.sqrt(r2)
            d = ng[:n] * np.sqrt(1 - r2)[:, None] + (bu[:n] * np.cos(r1)[:, None] + bv[:n] * np.sin(r1)[:, None]) * s[:, None]
            d = _normalize(d)
            sl = slice(filled, filled + n)
            out["origin"][sl] = origin32[:n]
            out["minT"][sl] = 1e-3
            out["dir"][sl] = d.astype(np.float32)
            out["maxT"][sl] = 1e6
            filled += n
            sample += 1
        batches.append(out)
    return batches


def random_rays(count, seed, extent=100.0, ymax=40.0):
    """Fully incoherent rays (uniform origins in the scene box, uniform directions);
    used by parity tests as a stress batch."""
    k = np.arange(count)
    o = np.stack([(hash_uniform(k, 51, seed) * 2 - 1) * extent,
                  hash_uniform(k, 52, seed) * ymax,
                  (hash_uniform(k, 53, seed) * 2 - 1) * extent], 1)
    z = hash_uniform(k, 54, seed).astype(np.float64) * 2 - 1
    ph = hash_uniform(k, 55, seed).astype(np.float64) * 2 * np.pi
    s = np.sqrt(np.maximum(0.0, 1 - z * z))
    rays = np.zeros(count, RAY_DTYPE)
    rays["origin"] = o.astype(np.float32)
    rays["dir"] = np.stack([s * np.cos(ph), z, s * np.sin(ph)], 1).astype(np.float32)
    rays["minT"] = 1e-3
    rays["maxT"] = 1e6
    return rays
