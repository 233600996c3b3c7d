"""Generates tests/golden/*: self-made golden vectors for the intersect-batch path.

The reference repository holds no golden vectors, known-answer tests or fixtures for this path
 and neither its CPU path (Embree binary absent) nor its own sources can be built in this
image, so these vectors come from oracle/racc_oracle.c (the CPU restatement), cross-checked here against
the double-precision brute-force arbiter before they are written.  They protect against regressions;
they do NOT pin the oracle to the reference ("parity unpinned").

    python tools/make_golden.py [--xl]     # rewrites tests/golden/golden_small.{npz,json}, algorithmic_bytes.json (--xl: the XL entries of the quality tree too)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle import oracle as orc
from rayaccel_amd import synth
from helpers import assert_matches_arbiter

OUT = os.path.join(ROOT, "tests", "golden")


def small():
    sc = synth.battlefield_synth(grid=24, boxes=12, quads=40)
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    prim, _ = synth.primary_rays(sc["camera"], 128, 128)
    prim = prim[::8]                                                     # 2048 coherent rays
    hits = orc.traverse(blobs, prim)
    rays = np.concatenate([prim, synth.diffuse_bounce_rays(sc, prim, hits, 2048), synth.random_rays(1024, seed=3, ymax=30.0)])
    env = sc["env"][::8, ::8].copy()                                     # 64x32 probe
    res, nv, npairs, depth = orc.traverse(blobs, rays, env=env, counters=True)
    ties = assert_matches_arbiter(res, sc, rays)
    np.savez_compressed(os.path.join(OUT, "golden_small.npz"),
                        vertices=sc["vertices"], indices=sc["indices"], env=env,
                        nodes=blobs["nodes"].view(np.uint8), pairs=blobs["pairs"].view(np.uint8), remap=blobs["remap"],
                        rays=rays.view(np.uint8), results=res.view(np.uint8), nv=nv, np=npairs)
    meta = dict(scene=sc["name"], rays=len(rays), pair_count=blobs["pair_count"], hits=int((res["triangle"] != 0xFFFFFFFF).sum()),
                algorithmic_bytes=orc.algorithmic_bytes(res, nv, npairs), max_stack=int(depth.max()), arbiter_ties=ties,
                provenance="oracle/racc_oracle.c, checked against orc_brute_closest; reference holds no vectors")
    json.dump(meta, open(os.path.join(OUT, "golden_small.json"), "w"), indent=1)
    print(meta)


def full_bytes():
    """Algorithmic bytes of the bench batches on the full battlefield-synth scene."""
    sc = synth.battlefield_synth()
    blobs = orc.build_scene(sc["vertices"], sc["indices"])
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    res, nv, npairs, depth = orc.traverse(blobs, prim, counters=True)
    out = dict(scene=sc["name"], triangles=len(sc["indices"]), inner_nodes=len(blobs["nodes"]), pairs=blobs["pair_count"],
               formula="B = 48*N + 64*sum(Nv) + 48*sum(Np) + 4*hits, reference traversal order on the reference-format BVH",
               coherent_1M=dict(bytes=orc.algorithmic_bytes(res, nv, npairs), nv_mean=float(nv.mean()), np_mean=float(npairs.mean()),
                                hit_rate=float((res["triangle"] != 0xFFFFFFFF).mean()), max_stack=int(depth.max())))
    for s in range(2):
        b = synth.diffuse_bounce_rays(sc, prim, res, 1 << 20, first_sample=s)
        r2, nv2, np2, d2 = orc.traverse(blobs, b, counters=True)
        out["diffuse_1M_sample%d" % s] = dict(bytes=orc.algorithmic_bytes(r2, nv2, np2), nv_mean=float(nv2.mean()), np_mean=float(np2.mean()),
                                             hit_rate=float((r2["triangle"] != 0xFFFFFFFF).mean()), max_stack=int(d2.max()))
    path = os.path.join(OUT, "algorithmic_bytes.json")
    try:        # sections other functions maintain (the XL entries; the quality trees) survive a rewrite
        old = json.load(open(path))
        for k in ("xl_1M", "xl_diffuse_1M", "quality1", "quality2"):
            if k in old:
                out[k] = old[k]
    except (OSError, ValueError):
        pass
    json.dump(out, open(path, "w"), indent=1)
    print(out)


def quality_bytes(quality=1, xl=False):
    """The same figures on the blobs of racc_host_scene_build_ex(quality): the PRODUCT's builder makes these trees (the reference has
    no such mode, the oracle restates none) — the oracle traverses them, in the reference's order, and counts."""
    import rayaccel_amd as ra
    path = os.path.join(OUT, "algorithmic_bytes.json")
    out = json.load(open(path))
    sec = out.setdefault("quality%d" % quality, {})
    sc = synth.battlefield_synth()
    host = ra.HostScene(sc["vertices"], sc["indices"], quality=quality)
    base = ra.HostScene(sc["vertices"], sc["indices"])
    prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    res0 = orc.traverse(base.blobs(), prim, threads=8)          # the bench bounces off the primaries' hits; any tree finds the same ones
    res, nv, npairs, depth = orc.traverse(host.blobs(), prim, counters=True, threads=8)
    sec.update(scene=sc["name"], inner_nodes=len(host.nodes), pairs=host.pair_count,
               what="racc_host_build_options.quality = %d (rayaccel_amd/csrc/scene_build.cpp): spatial splits + one pair per leaf + re-inserted subtrees; reference format, reference traversal order" % quality,
               coherent_1M=dict(bytes=orc.algorithmic_bytes(res, nv, npairs), nv_mean=float(nv.mean()), np_mean=float(npairs.mean()),
                                hit_rate=float((res["triangle"] != 0xFFFFFFFF).mean()), max_stack=int(depth.max())))
    for s, b in enumerate(synth.diffuse_bounce_batches(sc, prim, res, 1 << 20, range(2))):
        r2, nv2, np2, d2 = orc.traverse(host.blobs(), b, counters=True, threads=8)
        sec["diffuse_1M_sample%d" % s] = dict(bytes=orc.algorithmic_bytes(r2, nv2, np2), nv_mean=float(nv2.mean()), np_mean=float(np2.mean()),
                                              hit_rate=float((r2["triangle"] != 0xFFFFFFFF).mean()), max_stack=int(d2.max()))
    del res0
    if xl:
        sx = synth.battlefield_synth_xl()
        hx = ra.HostScene(sx["vertices"], sx["indices"], quality=quality)
        hits = orc.traverse(hx.blobs(), prim, threads=8)
        for key, rays in (("xl_1M", synth.random_rays(1 << 20, 7)), ("xl_diffuse_1M", synth.diffuse_bounce_rays(sx, prim, hits, 1 << 20))):
            r2, nv2, np2, d2 = orc.traverse(hx.blobs(), rays, counters=True, threads=8)
            sec[key] = dict(bytes=orc.algorithmic_bytes(r2, nv2, np2), nv_mean=float(nv2.mean()), np_mean=float(np2.mean()),
                            hit_rate=float((r2["triangle"] != 0xFFFFFFFF).mean()), max_stack=int(d2.max()),
                            inner_nodes=len(hx.nodes), pairs=hx.pair_count, scene=sx["name"])
    json.dump(out, open(path, "w"), indent=1)
    print(sec)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    small()
    if "--no-full" not in sys.argv:
        full_bytes()
        quality_bytes(1, xl="--xl" in sys.argv)
