"""Would XCD-local ray distribution pay?  Upper bound without touching the kernel: the same kind of rays, but ALL of them from one eighth of
   the image (a band of 128 of the 1024 pixel rows, 32 diffuse samples per pixel), so every XCD's L2 sees what it would see if the eight XCDs
   each worked on their own eighth.  Against the bench's kind of batch (the whole image, 4 samples per pixel), 4M rays each, one launch at a time;
   the oracle counts node visits and pair tests so that the comparison is per visit.   python tools/gpu_locality.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc

sc = synth.battlefield_synth()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=1)
prim, _ = synth.primary_rays(sc["camera"], 1024, 1024)
ref = orc.traverse(host.blobs(), prim, threads=16)
n = 1 << 22
batches = {"whole_image_4spp": synth.diffuse_bounce_rays(sc, prim, ref, n)}
for name, rows in (("band_rows_448_576_32spp", (448, 576)), ("band_rows_640_768_32spp", (640, 768)), ("band_rows_896_1024_32spp", (896, 1024))):
    sel = slice(rows[0] * 1024, rows[1] * 1024)
    if (ref["triangle"][sel] != 0xFFFFFFFF).sum() < 1000: continue
    batches[name] = synth.diffuse_bounce_rays(sc, prim[sel], ref[sel], n)
with ra.Context(device=0) as ctx:
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])
    for name, rays in batches.items():
        _, nvs, nps, _ = orc.traverse(host.blobs(), rays, threads=16, counters=True)
        nv, npair = float(nvs.mean()), float(nps.mean())
        d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(rays)
        ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 3)
        ms = float(np.median(ctx.intersect_device_timed(scene, env, d_r.ptr, d_o.ptr, n, 20)))
        print(json.dumps(dict(batch=name, ms_4M=round(ms, 4), mrays_per_s=round(n / ms / 1e3, 1), visits_per_ray=round(nv, 2), pair_tests_per_ray=round(npair, 2), ps_per_visit=round(ms * 1e9 / (n * nv), 3))), flush=True)
        d_r.free(); d_o.free()
    scene.destroy(); env.destroy()
