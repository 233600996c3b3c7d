"""Soak test: random batch sizes / launch options / kernel variants / lanes against the oracle, bit for bit.
   python tools/gpu_fuzz.py [rounds] [seed]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra
from rayaccel_amd import synth
from oracle import oracle as orc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import QUANT_VARIANTS, WIDE_VARIANTS, assert_bit_exact, assert_same_closest_hit        # hits bit for bit (wide kernels: up to exact-distance ties), miss colours (acosf: libm vs ocml) to 1e-5

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
# RACC_FUZZ_SCENE=soup-synth|city-synth: another scene class at reduced size (soup: leaves of a dozen pairs, stacks deeper than the LDS part)
sc = {"soup-synth": lambda: synth.soup_synth(triangles=60000, clusters=24), "city-synth": lambda: synth.city_synth(blocks=22)}.get(
    os.environ.get("RACC_FUZZ_SCENE", ""), lambda: synth.battlefield_synth(grid=160, boxes=900, quads=4000))()
host = ra.HostScene(sc["vertices"], sc["indices"], quality=int(os.environ.get("RACC_FUZZ_QUALITY", "1")))      # the quality tree by default: the oracle traverses the same blobs
blobs = host.blobs()
prim, _ = synth.primary_rays(sc["camera"], 512, 512)
ref_prim = orc.traverse(blobs, prim, env=sc["env"], threads=8)
pool = np.concatenate([prim] + [synth.diffuse_bounce_rays(sc, prim, ref_prim, 1 << 18, first_sample=s) for s in range(2)])
pool = pool[rng.permutation(len(pool))]
ref_pool = orc.traverse(blobs, pool, env=sc["env"], threads=8)
bad = 0
for rnd in range(rounds):
    opt = dict(kernel_variant=int(rng.choice([0, 0, 0, 41, 43, 45, 46, 49, 50, 50, 50, 51, 53, 60, 61, 62, 63, 70, 70] + [v for v in (22, 23, 24, 17, 11, 1, 31, 34, 38) if v in ra.engine.available_variants()])), lanes=int(rng.integers(1, 5)), chain_min_rays=int(rng.choice([0, 1, 1, 5000])), chain_launches=int(rng.choice([0, 0, 0, 3, 2])))      # (0: lazy chain, 3: per-launch kernels, 2: off)
    if rng.random() < 0.6:
        opt.update(waves_per_simd=int(rng.integers(1, 9)), refill_min=int(rng.integers(1, 65)), leaf_min=int(rng.integers(1, 65)),
                   chunk=int(rng.choice([1, 7, 32, 64, 100, 128, 1000])), tail_active=int(rng.integers(1, 70)),
                   thin_reps=int(rng.integers(1, 20)), inner_reps=int(rng.integers(1, 9)), coop_same_pct=int(rng.choice([0, 1, 25, 100, 101])), leaf_step=int(rng.choice([0, 0, 2, 3])), drain_prefetch=int(rng.choice([0, 1, 3])))
    with ra.Context(device=0, **opt) as ctx:
        scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
        env = ctx.create_environment(sc["env"])
        errs = []

        chained = bool(rng.random() < 0.4)
        host_async = (not chained) and bool(rng.random() < 0.4)      # page-locked host batches enqueued without waiting (racc_hip_intersect_async): the pipeline across batches

        def check(got, off, n, rays, lane):
            v = opt["kernel_variant"]
            try:
                if v in QUANT_VARIANTS:
                    assert_same_closest_hit(got, ref_pool[off:off + n], arbiter=dict(vertices=sc["vertices"], indices=sc["indices"], rays=rays))
                elif v in WIDE_VARIANTS:
                    assert_same_closest_hit(got, ref_pool[off:off + n])
                else:
                    assert_bit_exact(got, ref_pool[off:off + n])
            except AssertionError as e:
                errs.append((lane, n, off, str(e)[:160]))

        def work(lane, seed):
            r2 = np.random.default_rng(seed)
            pending = []
            for _ in range(4):                                      # back-to-back launches of different sizes on one lane
                n = int(r2.choice([1, 2, 63, 64, 65, 255, 4097, int(r2.integers(1, 1 << 14)), int(r2.integers(1, len(pool) - 1))]))
                off = int(r2.integers(0, len(pool) - n + 1))
                rays = np.ascontiguousarray(pool[off:off + n])
                if chained:                                         # device-resident, engine's own streams: chained launches, lanes rotated
                    d_r = ctx.alloc(n * 32); d_o = ctx.alloc(n * 16); d_r.upload(rays)
                    ctx.intersect_device(scene, env, d_r.ptr, d_o.ptr, n, lane=ra.LANE_AUTO)
                    pending.append((d_r, d_o, n, off, rays))
                    continue
                if host_async:
                    rays = rays.copy()          # (own memory: page-locking overlapping slices of the pool from several threads is not what is under test)
                    out = np.zeros(n, ra.RESULT_DTYPE)
                    toks = [ctx.register_host(rays), ctx.register_host(out)]
                    ctx.intersect_async(scene, env, rays, out, lane=lane)
                    pending.append((toks, out, n, off, rays))
                    continue
                got = ctx.intersect(scene, env, rays, lane=lane)
                check(got, off, n, rays, lane)
            if host_async:
                ctx.wait(lane)
                for toks, out, n, off, rays in pending:
                    check(out, off, n, rays, lane)
                    for t in toks: ctx.unregister_host(t)
                return
            if chained:
                ctx.wait(ra.LANE_AUTO)
                for d_r, d_o, n, off, rays in pending:
                    check(d_o.download(orc.RESULT_DTYPE, n), off, n, rays, lane)
                    d_r.free(); d_o.free()
        ts = [threading.Thread(target=work, args=(l, int(rng.integers(1 << 30)))) for l in range(opt["lanes"])]
        for t in ts: t.start()
        for t in ts: t.join()
        if errs:
            bad += 1
            print("MISMATCH", opt, errs[:3], flush=True)
        scene.destroy(); env.destroy()
print("fuzz: %d rounds, %d with mismatches" % (rounds, bad))
sys.exit(1 if bad else 0)
