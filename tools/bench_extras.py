"""The untimed parts of bench.py: everything the JSON line carries beside `value` — the oracle check of every timed record and the CPU legs,
the roofline object, the all-gather figures, the single-GPU extras (other configs, host-buffer paths, the reference builder's tree, the
compressed 4-wide kernel, the path tracer), battlefield-synth-XL.  Split out of bench.py in round 6 (no behaviour of the timed region lives
here).  Every function takes the state object `S` bench.py fills (context, scene, arrays, arguments) and returns plain dicts.

Only this module and bench.py's own `cpu_baseline` leg touch oracle/ — as the checker and the CPU leg, never on the product path."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED_GBS = 6290.0      # same guide: float4-copy ceiling
L2_PEAK_GBS = 34500.0          # same guide, "L2 (per XCD)": 4 MiB x 8, ~34.5 TB/s aggregate
PCIE_GBS_PER_DIRECTION = 56.0  # page-locked copies on the GPU boxes, one direction alone (tools/microbench/pcie.hip; 49 + 49 with both at once)
CU_CLOCK_HZ, CUS = 2.4e9, 256
RAYS_PER_BATCH = 1 << 20
XL_RAY_SEED = 7


def gather_peak_gbs(ceiling):
    """The CU gather path's measured ceiling (bytes per clock per CU, tools/microbench/gather64.hip) as a whole-chip rate in GB/s."""
    return ceiling * CUS * CU_CLOCK_HZ / 1e9


def roofline_core(alg, ms, traffic, ceiling, hbm_binds=False):
    """The measurement contract's roofline fields for one (kernel, batch): `achieved` = algorithmic bytes per launch (oracle
    counters) over the kernel's launch duration.  `bound` / `peak` / `frac` name what the bytes are held against:
      * a scene that lives in the L2s and the Infinity Cache (battlefield-synth, 66 MB): HBM cannot bind — the algorithmic bytes exceed what
        8 TB/s could deliver (`hbm_algorithmic_frac` > 1) while a fourteenth of them cross the fabric (`hbm_traffic_frac`).  They pass through
        the CU's vector-memory gather path, whose ceiling for random 64-byte records is MEASURED on this part (gather64, mode 2): that is the
        roof, `frac` is < 1 against it;
      * battlefield-synth-XL (1.5 GB, L2 hit rate 0.53): `bound` = "hbm", `frac` against the 8 TB/s peak, `traffic` = 0.9 x the algorithmic bytes.
    `traffic` = fabric (L2-miss) bytes per launch from the committed rocprofv3 --pmc passes (2 x FETCH_SIZE + WRITE_SIZE), or None."""
    if not alg or not ms:
        return None
    achieved = alg / (ms * 1e-3) / 1e9
    hbm_alg = achieved / HBM_PEAK_GBS
    hbm_traffic = traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None
    if hbm_binds or not ceiling:
        bound, peak = "hbm", HBM_PEAK_GBS
    else:
        bound, peak = "cu_gather_path", gather_peak_gbs(ceiling)
    out = {"bound": bound, "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "GB/s", "frac": round(achieved / peak, 4), "traffic": traffic,
           "algorithmic_bytes_per_launch": int(alg), "kernel_ms_avg": round(ms, 4),
           "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_algorithmic_frac": round(hbm_alg, 4), "hbm_traffic_frac": round(hbm_traffic, 4) if hbm_traffic is not None else None,
           "hbm_traffic_frac_of_measured_ceiling": round(traffic / (ms * 1e-3) / 1e9 / HBM_MEASURED_GBS, 4) if traffic else None,
           "traffic_over_algorithmic": round(traffic / alg, 4) if traffic else None}
    if bound == "hbm" and not hbm_binds:
        out["bound_note"] = "no gather-path ceiling committed (profiles/<round>/microbench.json missing): held against the HBM peak, which does not bind a cache-resident scene"
    return out


def same_records(got, ref, colours=True):
    """primId, t, u, v bit for bit; miss colours to 1e-5 (acosf of ocml vs libm)."""
    hit = ref["triangle"] != 0xFFFFFFFF
    if not np.array_equal(got["triangle"], ref["triangle"]):
        return False
    if any(not np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)) for f in ("t", "u", "v")):
        return False
    return not colours or all(np.allclose(got[f][~hit], ref[f][~hit], rtol=1e-5, atol=1e-5) for f in ("t", "u", "v"))


# ------------------------------------------------------------------------------------------------------------ the all-gather
def allgather_extras(S):
    """BASELINE configs[3]'s exchange step: every rank's Result shard gathered on every GPU through the C-ABI's own entry
    (racc_hip_allgather_results -> ncclAllGather over xGMI, one message per rank).  Two figures for the same K steps:
      serialised   trace, wait, gather, per step (round 5's figure);
      overlapped   double-buffered: step k is traced on one of two caller streams (racc_hip_intersect_device with a stream: a stand-alone
                   launch, stream-ordered) and its gather is issued behind it ON THAT STREAM, while step k+1 traces on the other stream —
                   SURVEY §8(e)'s cost model has the gather (0.11 ms direct, 0.77 ms ring per 16 MiB shard) comparable with the trace.
    Runs on every rank (a collective); at N = 1 it is RCCL with one rank: the plumbing, no fabric."""
    import torch
    import torch.distributed as dist
    import rayaccel_amd as ra
    ctx, n, world, rank, steps = S.ctx, S.n, S.world, S.rank, S.args.steps
    uid = [ra.Comm.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    comm = ra.Comm(ctx, uid[0], rank, world)
    per = -(-S.total_rays // world) if S.args.mode == "strong" else n
    out = {}
    try:
        gathered = [torch.empty((world * per, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        send = [torch.zeros((per, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        comm.allgather_results(send[0].data_ptr(), gathered[0].data_ptr(), per)
        torch.cuda.synchronize(); S.barrier()

        def max_over_ranks(dt):
            if world == 1:
                return dt
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        # serialised
        t1 = time.perf_counter()
        for k in range(steps):
            lane = k % S.lanes
            ctx.intersect_device(S.scene, S.env, S.d_sets[k % len(S.d_sets)].data_ptr(), send[0].data_ptr(), n, lane=lane)
            ctx.wait(lane)
            comm.allgather_results(send[0].data_ptr(), gathered[0].data_ptr(), per)
        ctx.synchronize(); torch.cuda.synchronize()
        dt = max_over_ranks(time.perf_counter() - t1)
        out["with_allgather_of_results"] = {"mrays_per_s": round(S.total_rays * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 4),
                                            "bytes_gathered_per_step": int(gathered[0].numel() * 4),
                                            "how": "racc_hip_allgather_results (C-ABI -> ncclAllGather), one message per rank; trace, wait, gather serialised per step"}
        # overlapped
        streams = [ctx.create_stream() for _ in range(2)]
        S.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(steps):
            b = k & 1
            ctx.intersect_device(S.scene, S.env, S.d_sets[k % len(S.d_sets)].data_ptr(), send[b].data_ptr(), n, lane=b, stream=streams[b])
            comm.allgather_results(send[b].data_ptr(), gathered[b].data_ptr(), per, stream=streams[b])
        for s in streams:
            ctx.stream_synchronize(s)
        torch.cuda.synchronize()
        dt2 = max_over_ranks(time.perf_counter() - t1)
        ctx.synchronize()      # (racc_hip_synchronize also tells the engine that no lane has a launch in flight any more: launches on a caller's stream are
                               #  not waited for through racc_hip_wait, and the next lone launch would otherwise take the thin grid of an overlapped one)
        ok = bool(torch.equal(gathered[(steps - 1) & 1][rank * per: rank * per + n].view(torch.int32), S.outs[(steps - 1) % len(S.d_sets)].view(torch.int32))) if S.args.mode == "weak" and steps <= len(S.outs) else None
        out["with_allgather_of_results_overlapped"] = {
            "mrays_per_s": round(S.total_rays * steps / dt2 / 1e6, 1), "ms_per_step": round(dt2 / steps * 1e3, 4), "bytes_gathered_per_step": int(gathered[0].numel() * 4),
            "ranks": world, "own_shard_in_the_gathered_array_equals_the_timed_results": ok,
            "how": "two caller streams in turn: step k's stand-alone traversal launch and, behind it on the same stream, its ncclAllGather; step k+1 traces on the "
                   "other stream meanwhile (send and receive arrays double-buffered).  At N = 1 the collective has one rank: plumbing only, no xGMI traffic"}
        for s in streams:
            ctx.destroy_stream(s)
    finally:
        comm.destroy()
    return out


# ------------------------------------------------------------------------------------------------------------ single-GPU extras
def single_gpu_extras(S):
    import torch
    import rayaccel_amd as ra
    from rayaccel_amd import synth
    ctx, scene, env, n, args, sc, host = S.ctx, S.scene, S.env, S.n, S.args, S.sc, S.host
    outs, d_sets, d_rays, primary = S.outs, S.d_sets, S.d_sets[0], S.primary
    extras = {}

    def timed_serial(rays_t, out_t, iters):
        return ctx.intersect_device_timed(scene, env, rays_t.data_ptr(), out_t.data_ptr(), rays_t.shape[0], iters)
    # one launch at a time (what round 1 reported as `value`): the launch's drain is exposed
    timed_serial(d_rays, outs[1], 2)
    t1 = time.perf_counter()
    serial_ms = timed_serial(d_rays, outs[1], args.steps)
    extras["one_launch_at_a_time"] = {"mrays_per_s": round(args.steps * n / (time.perf_counter() - t1) / 1e6, 1), "kernel_ms_avg": round(float(np.mean(serial_ms)), 4)}
    if not torch.equal(outs[1].view(torch.int32), S.d_ref_bits):   # bit compare (a miss id reads as NaN in f32)
        sys.exit("bench: overlapped launches changed the results")
    d_prim = torch.from_numpy(primary.view(np.float32).reshape(len(primary), 8).copy()).cuda()
    d_prim_out = torch.zeros((len(primary), 4), dtype=torch.float32, device="cuda")
    timed_serial(d_prim, d_prim_out, 2)
    pm = float(np.median(timed_serial(d_prim, d_prim_out, 10)))
    extras["coherent_1M"] = {"ms_per_step": round(pm, 4), "mrays_per_s": round(len(primary) / pm / 1e3, 1)}
    for k in range(args.warmup + args.steps):           # configs[1] issued like the timed region: chained, lanes rotated
        if k == args.warmup:
            ctx.wait(ra.LANE_AUTO); torch.cuda.synchronize(); t1 = time.perf_counter()
        ctx.intersect_device(scene, env, d_prim.data_ptr(), outs[k % len(outs)].data_ptr(), len(primary), lane=ra.LANE_AUTO)
    ctx.wait(ra.LANE_AUTO); torch.cuda.synchronize()
    extras["coherent_1M"]["back_to_back_mrays_per_s"] = round(len(primary) * args.steps / (time.perf_counter() - t1) / 1e6, 1)
    # PCIe-inclusive rate of the host-buffer entry points (never `value`): pageable arrays here, in this process ...
    res_host = np.zeros(n, ra.RESULT_DTYPE)
    ctx.intersect(scene, env, S.bounce, res_host)
    t1 = time.perf_counter()
    for _ in range(3):
        ctx.intersect(scene, env, S.bounce, res_host)
    extras["host_buffers_pcie_inclusive_mrays_per_s"] = round(3 * n / (time.perf_counter() - t1) / 1e6, 1)
    if not np.array_equal(res_host.view(np.uint32).reshape(-1, 4), S.d_ref_bits.cpu().numpy().view(np.uint32)):
        sys.exit("bench: the host-buffer path changed the results")
    # ... and page-locked arrays the way a host application binds the C-ABI: tools/host_path_bench.py in a process of its own, without torch
    # (torch ships its own, older HIP runtime; with it loaded the same pipeline moves 40-47 instead of 54 GB/s into the GPU: measured both ways in round 4)
    if S.standard_scene:
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_path_bench.py"), "--grid", str(args.grid), "--device", str(S.device),
                                "--link-gbs", str(PCIE_GBS_PER_DIRECTION)], capture_output=True, text=True, timeout=600, cwd=ROOT)
            extras["host_buffers_page_locked"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else {"error": (p.stderr or p.stdout)[-300:]}
        except Exception as e:   # noqa: BLE001
            extras["host_buffers_page_locked"] = {"error": str(e)[:200]}

    # The quality-0 tree (the reference's builder, byte-identical to the oracle's restatement of Bvh2.cpp) in the SAME loop as `value`,
    # same context, same rays, and one launch at a time: what the tree post-processing of the default scene build buys.
    if host.quality and args.mode == "weak" and not S.xl_run:
        h0 = ra.HostScene(sc["vertices"], sc["indices"], quality=0)
        scene0 = ctx.upload_scene(h0.nodes, h0.pairs, h0.remap)

        def run0(steps):
            for k in range(steps):
                ctx.intersect_device(scene0, env, d_sets[k % len(d_sets)].data_ptr(), outs[k % len(outs)].data_ptr(), n, lane=ra.LANE_AUTO)
            ctx.wait(ra.LANE_AUTO)
        iso0 = ctx.intersect_device_timed(scene0, env, d_rays.data_ptr(), outs[-1].data_ptr(), n, 30)
        if args.warmup:
            run0(args.warmup)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run0(args.steps)
        torch.cuda.synchronize()
        dt0 = time.perf_counter() - t1
        same_prim = int((outs[0].view(torch.int32)[:, 0] == S.d_ref_bits[:, 0]).sum().item())
        extras["reference_builder_tree"] = {
            "mrays_per_s_same_loop_as_value": round(n * args.steps / dt0 / 1e6, 1), "ms_per_step": round(dt0 / args.steps * 1e3, 4),
            "kernel_ms_avg_one_launch_at_a_time": round(float(np.mean(iso0[len(iso0) // 2:])), 4),
            "inner_nodes": len(h0.nodes), "pairs": int(h0.pair_count),
            "primIds_equal_to_the_quality_tree": "%d of %d" % (same_prim, n),
            "what": "racc_host_build_options.quality = 0: Bvh2.cpp:257-535 restated, byte-identical to the oracle's builder (what RACC_BUILD_QUALITY=0 makes racc::createScene "
                    "build); t/u/v of the two trees agree to rounding, primIds up to exact-distance ties (tests/test_quality_build.py, tests/test_gpu_quality.py)"}
        scene0.destroy()
        del h0

    # Batch-size scaling of the traversal kernel (same diffuse rays, 8 sample sets): T(N) = fixed + per-ray cost.
    if S.full and args.mode == "weak" and S.standard_scene:
        many = np.concatenate(S.ray_sets) if len(S.ray_sets) == 8 else np.concatenate(synth.diffuse_bounce_batches(sc, primary, S.primary_hits, RAYS_PER_BATCH, range(8)))
        d_many = torch.from_numpy(many.view(np.float32).reshape(len(many), 8).copy()).cuda()
        d_many_out = torch.zeros((len(many), 4), dtype=torch.float32, device="cuda")
        scaling = {}
        for nn in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 23):
            ctx.intersect_device_timed(scene, env, d_many.data_ptr(), d_many_out.data_ptr(), nn, 2)
            scaling[str(nn)] = round(float(np.median(ctx.intersect_device_timed(scene, env, d_many.data_ptr(), d_many_out.data_ptr(), nn, 7))), 4)
        slope = (scaling[str(1 << 23)] - scaling[str(1 << 20)]) / 7.0          # ms per 2^20 rays
        extras["batch_scaling"] = {"kernel_ms_by_rays": scaling, "steady_state_mrays_per_s": round((1 << 20) / slope / 1e3, 1), "fixed_ms": round(scaling[str(1 << 20)] - slope, 4)}
        # The compressed 4-wide kernel (kernel_variant 50 = racc::setFastTraversal; racc_kernel_v10.inc) on the same batches, its own context, same
        # scene blobs: single launches, and the SAME loop as the timed region — K lazily chained steps after W warm-up steps, the sample sets rotating.
        try:
            with ra.Context(device=S.device, kernel_variant=50, time_kernels=0) as wctx:
                wscene = wctx.upload_scene(host.nodes, host.pairs, host.remap)
                wenv = wctx.create_environment(sc["env"])
                wide = {}
                for nn in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 23):
                    wctx.intersect_device_timed(wscene, wenv, d_many.data_ptr(), d_many_out.data_ptr(), nn, 2)
                    wide[str(nn)] = round(float(np.median(wctx.intersect_device_timed(wscene, wenv, d_many.data_ptr(), d_many_out.data_ptr(), nn, 7))), 4)
                wslope = (wide[str(1 << 23)] - wide[str(1 << 20)]) / 7.0

                def wrun(steps):
                    for k in range(steps):
                        wctx.intersect_device(wscene, wenv, d_sets[k % len(d_sets)].data_ptr(), outs[k % len(outs)].data_ptr(), n, lane=ra.LANE_AUTO)
                    wctx.wait(ra.LANE_AUTO)
                if args.warmup:
                    wrun(args.warmup)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                wrun(args.steps)
                torch.cuda.synchronize()
                wdt = time.perf_counter() - t1
                differing = int((outs[0].view(torch.int32) != S.d_ref_bits).any(dim=1).sum().item())
                hits_differing = int(((outs[0].view(torch.int32) != S.d_ref_bits).any(dim=1) & (S.d_ref_bits[:, 0] != -1)).sum().item())
                extras["compressed_wide_kernel_variant_50"] = {
                    "mrays_per_s_same_loop_as_value": round(n * args.steps / wdt / 1e6, 1), "ms_per_step": round(wdt / args.steps * 1e3, 4),
                    "loop": "lazily chained, %d sample sets in rotation: exactly the timed region's loop (round 6: the kernel has a lazy-chain instantiation)" % len(d_sets),
                    "kernel_ms_by_rays": wide, "default_kernel_ms_by_rays": {k: scaling[k] for k in wide},
                    "steady_state_mrays_per_s": round((1 << 20) / wslope / 1e3, 1), "fixed_ms": round(wide[str(1 << 20)] - wslope, 4),
                    "records_differing_from_default_in_1M": differing, "hit_records_differing_from_default_in_1M": hits_differing,
                    "note": "opt-in fast mode (racc_hip_options::kernel_variant 50, racc::setFastTraversal): 64 B 4-wide nodes, child boxes quantised conservatively to 8 bits: half the "
                            "node bytes and vector-memory instructions per ray; same closest hit as the default kernel except exact-distance ties and arbiter-confirmed closer hits "
                            "(miss colours differ in the last bit: another instruction order of the same lookup)"}
                wscene.destroy(); wenv.destroy()
        except ra.RaccError as e:
            extras["compressed_wide_kernel_variant_50"] = {"error": str(e)}
        del d_many, d_many_out

    # BASELINE configs[4]: the path tracer, 1920x1080, end to end on this GPU.  Device-resident consumer at 64 spp; the reference-shaped consumer
    # (spawn/shade callbacks on host threads through racc::render, PCIe both ways) at 8 spp.  Both render the same image (tests/test_gpu_pathtracer.py).
    if S.full and args.mode == "weak" and S.standard_scene:
        import tempfile
        from rayaccel_amd.engine import path_trace
        tmp = tempfile.NamedTemporaryFile(suffix=".bin", delete=False)
        tmp.close()
        saved = os.environ.get("RACC_BUILD_QUALITY")
        try:
            synth.write_scene_bin(tmp.name, sc, viewport=(1920, 1080))
            if args.quality is not None:      # (the consumers build their own scene through racc_host_scene_build: the library default unless the line was asked for another tree)
                os.environ["RACC_BUILD_QUALITY"] = str(args.quality)
            _, sg = path_trace(tmp.name, 1920, 1080, 0, 64, device=S.device, shading="gpu")
            _, sh = path_trace(tmp.name, 1920, 1080, 0, 8, device=S.device, shading="cpu", cpu_threads=S.usable_cores())
            try:
                pr = subprocess.run([os.path.join(ROOT, "tests", "cpp", "render_check"), tmp.name, "--null-callbacks", "1920", "1080", "16", "4"],
                                    capture_output=True, text=True, timeout=300, env=dict(os.environ, RACC_CPU_THREADS=str(S.usable_cores())))
                sched = json.loads(pr.stdout.strip().splitlines()[-1]) if pr.returncode == 0 else {"error": (pr.stderr or pr.stdout)[-300:]}
            except Exception as e:   # noqa: BLE001
                sched = {"error": str(e)[:200]}
            extras["path_tracer_1080p"] = {
                "scheduler_only_null_callbacks": sched,
                "gpu_shading_64spp": {"mrays_per_s": round(sg["rays_traced"] / sg["seconds"] / 1e6, 1), "seconds": round(sg["seconds"], 4), "rays": int(sg["rays_traced"])},
                "host_shading_8spp": {"mrays_per_s": round(sh["rays_traced"] / sh["seconds"] / 1e6, 1), "seconds": round(sh["seconds"], 4), "rays": int(sh["rays_traced"]),
                                      "shade_threads": int(sh["threads"])}}
        finally:
            os.unlink(tmp.name)
            if saved is None:
                os.environ.pop("RACC_BUILD_QUALITY", None)
            else:
                os.environ["RACC_BUILD_QUALITY"] = saved
    return extras


# ------------------------------------------------------------------------------------------------------------ battlefield-synth-XL
def xl_rooflines(S):
    """battlefield-synth-XL: the regime in which HBM can bind (rank 0, N = 1)."""
    import torch
    import rayaccel_amd as ra
    from rayaccel_amd import synth
    ctx, args = S.ctx, S.args
    xl = {}
    prof_x = S.committed_profile() or {}
    sx = synth.battlefield_synth_xl()
    hx = ra.HostScene(sx["vertices"], sx["indices"], quality=args.quality)
    scene_x = ctx.upload_scene(hx.nodes, hx.pairs, hx.remap)
    env_x = ctx.create_environment(sx["env"])
    hits_x = ctx.intersect(scene_x, env_x, S.primary)
    xl_batches = (("xl", "1M incoherent rays (origins and directions uniform over the scene)", synth.random_rays(RAYS_PER_BATCH, XL_RAY_SEED)),
                  ("xl_diffuse", "1M first-bounce diffuse rays of the bench camera", synth.diffuse_bounce_rays(sx, S.primary, hits_x, RAYS_PER_BATCH)))
    for key, what, rays_x in xl_batches:
        d_rx = torch.from_numpy(rays_x.view(np.float32).reshape(len(rays_x), 8).copy()).cuda()
        d_ox = torch.zeros((len(rays_x), 4), dtype=torch.float32, device="cuda")
        ms_x = ctx.intersect_device_timed(scene_x, env_x, d_rx.data_ptr(), d_ox.data_ptr(), len(rays_x), 40)
        ms_x = float(np.mean(ms_x[len(ms_x) // 2:]))
        alg_x, src_x = None, None
        if not args.no_cpu_baseline:
            from oracle import oracle            # checker only
            ref_x, nv_x, np_x, _ = oracle.traverse(hx.blobs(), rays_x, env=sx["env"], counters=True, threads=S.usable_cores())
            alg_x, src_x = oracle.algorithmic_bytes(ref_x, nv_x, np_x), "oracle counters, live"
            if not same_records(d_ox.cpu().numpy().view(ra.RESULT_DTYPE).reshape(-1), ref_x, colours=False):
                sys.exit("bench: GPU results on battlefield-synth-XL (%s) differ from the oracle — refusing to report a number" % key)
        else:
            try:
                with open(os.path.join(ROOT, "tests", "golden", "algorithmic_bytes.json")) as f:
                    gx = json.load(f)
                    alg_x, src_x = (gx["quality%d" % hx.quality] if hx.quality else gx)[key + "_1M"]["bytes"], "tests/golden/algorithmic_bytes.json"
            except (OSError, KeyError, ValueError):
                pass
        px = prof_x.get(key, {})
        # HBM binds where most of the algorithmic bytes really cross the fabric (the incoherent batch: 0.88 of them); the camera's first-bounce rays
        # touch a small part of the 1.5 GB (a tenth of their bytes cross it): held to the gather path like the cache-resident scene
        tx = px.get("fabric_bytes_per_launch")
        r = roofline_core(alg_x, ms_x, tx, S.gather_ceiling(), hbm_binds=bool(tx and alg_x and tx > 0.5 * alg_x) or (key == "xl" and not tx)) or {"kernel_ms_avg": round(ms_x, 4)}
        r.update({"workload": "battlefield-synth-XL, %d triangles, %s" % (len(sx["indices"]), what), "mrays_per_s": round(len(rays_x) / ms_x / 1e3, 1),
                  "algorithmic_source": src_x, "device_bytes": int(scene_x.info["device_bytes"]),
                  "limiter": {q: px.get(q) for q in ("td_busy_frac", "ta_busy_frac", "valu_busy_frac", "l2_hit_rate", "kernel_ms_isolated", "fetch_bytes_per_launch", "write_bytes_per_launch")} if px else None})
        xl[key] = r
        del d_rx, d_ox
    scene_x.destroy(); env_x.destroy()
    return xl


# ------------------------------------------------------------------------------------------------------------ oracle check + CPU legs
def oracle_check_and_cpu_legs(S):
    """Every record of every sample set of the timed rotation against the oracle (refusing to report a number otherwise), the algorithmic
    bytes of SURVEY §8(d) from the oracle's counters on the blobs the GPU traversed, and the CPU legs: the scalar port on all usable host
    cores, the 8-wide AVX2 form of the same traversal where the oracle has one, a system Embree where a box has one, and the reference's own
    OpenCL kernel on this GPU.  N = 1, rank 0 only."""
    import rayaccel_amd as ra
    from oracle import oracle            # checker / CPU leg only; never on the product path
    args, sc, n, bounce = S.args, S.sc, S.n, S.bounce
    extras = {}
    blobs = S.host.blobs()
    threads = S.usable_cores()
    ref, nv, npairs, _ = oracle.traverse(blobs, bounce, env=sc["env"], counters=True, threads=threads)
    alg_bytes = oracle.algorithmic_bytes(ref, nv, npairs)
    src = "oracle counters, live, on the blobs the GPU traverses (racc_host_build_options.quality = %d)" % S.host.quality
    alg_by_set = [alg_bytes]
    visits = {"node_visits_per_ray": round(float(nv.mean()), 2), "pair_tests_per_ray": round(float(npairs.mean()), 2), "hit_rate": round(float((ref["triangle"] != 0xFFFFFFFF).mean()), 4)}
    for k, bits in enumerate(S.set_bits, 1):      # the other sample sets of the rotation: every record against the oracle as well
        ref_k, nv_k, np_k, _ = oracle.traverse(blobs, S.ray_sets[k], env=sc["env"], counters=True, threads=threads)
        if not same_records(bits.cpu().numpy().view(ra.RESULT_DTYPE).reshape(-1), ref_k, colours=False):
            sys.exit("bench: GPU results of sample set %d differ from the oracle — refusing to report a number" % k)
        alg_by_set.append(oracle.algorithmic_bytes(ref_k, nv_k, np_k))
    if not same_records(S.d_ref_bits.cpu().numpy().view(ra.RESULT_DTYPE).reshape(-1), ref):
        sys.exit("bench: GPU results (primId, t, u, v bits; miss colours to 1e-5) differ from the oracle — refusing to report a number")

    def time_leg(fn):
        cpu_out = np.zeros(n, oracle.RESULT_DTYPE)
        fn(cpu_out, 1)                                   # warm-up: faults pages, starts clocks
        t1 = time.perf_counter()
        fn(cpu_out, 1)
        one = time.perf_counter() - t1
        repeat = int(min(64, max(2, 6.0 / max(one, 1e-3))))                                 # ~6 s of CPU work per timing
        times = []
        for _ in range(3):
            t1 = time.perf_counter()
            fn(cpu_out, repeat)
            times.append((time.perf_counter() - t1) / repeat)
        return float(np.median(times)), repeat, cpu_out
    t_scalar, rep_s, out_s = time_leg(lambda o, r: oracle.traverse(blobs, bounce, env=sc["env"], threads=threads, repeat=r, out=o))
    cpu_baseline = {"value": round(n / t_scalar / 1e6, 2), "unit": "Mrays/s", "cores": threads, "kind": "port",
                    "sample": "the full 1,048,576-ray batch of the timed workload, %d passes per timing x 3 timings (median), %d pthreads x 1024-ray slices; SCALAR BVH2 port of "
                              "the reference's traversal, not Embree-class (the reference's CPU path is binary-only Embree 2.x bvh8/AVX2, unavailable here)" % (rep_s, threads)}
    if hasattr(oracle, "traverse_simd") and oracle.simd_available():
        # The reference hands its CPU leg 8 rays at a time (Scene.cpp:386-428, rtcIntersect8, isa=avx2): an 8-wide AVX2 form of the oracle's
        # traversal — one packet of 8 rays walks the BVH2 together, lane masks, same arithmetic per lane — bit-identical to the scalar port
        # (tests/test_oracle.py).  Still not Embree (no bvh8, no triangle4 leaves, no spatial reordering): a port, of the SIMD kind.
        t_simd, rep_v, out_v = time_leg(lambda o, r: oracle.traverse_simd(blobs, bounce, env=sc["env"], threads=threads, repeat=r, out=o))
        if not np.array_equal(out_v.view(np.uint8), out_s.view(np.uint8)):
            sys.exit("bench: the 8-wide CPU port and the scalar port disagree")
        isa, by_width = "avx2", {"avx2_8_lanes": round(n / t_simd / 1e6, 2)}
        if oracle.simd_width() == 16:       # the same again with 16 rays per zmm register where the host has AVX-512 (the GPU boxes' EPYC 9575F does)
            t_16, rep_16, out_16 = time_leg(lambda o, r: oracle.traverse_simd(blobs, bounce, env=sc["env"], threads=threads, repeat=r, out=o, width=16))
            if not np.array_equal(out_16.view(np.uint8), out_s.view(np.uint8)):
                sys.exit("bench: the 16-wide CPU port and the scalar port disagree")
            by_width["avx512_16_lanes"] = round(n / t_16 / 1e6, 2)
            if t_16 < t_simd: isa, t_simd, rep_v = "avx512", t_16, rep_16
        cpu_baseline = dict(cpu_baseline, value=round(n / t_simd / 1e6, 2), kind="simd-port", isa=isa, mrays_per_s_by_width=by_width,
                            scalar_port_mrays_per_s=cpu_baseline["value"], simd_over_scalar=round(t_scalar / t_simd, 2),
                            sample="the full 1,048,576-ray batch of the timed workload, %d passes per timing x 3 timings (median), %d pthreads x 1024-ray slices; SIMD form "
                                   "of the oracle's BVH2 traversal (one ray per vector lane, 8 as Scene.cpp:386-428 hands them to rtcIntersect8 or 16 where the host has AVX-512: "
                                   "`value` is the faster, `mrays_per_s_by_width` both), bit-identical to the scalar port "
                                   "(`scalar_port_mrays_per_s`); NOT Embree: the reference's CPU path is binary-only Embree 2.x bvh8/AVX2, unavailable here" % (rep_v, threads))
    try:        # optional row: a system Embree through oracle/embree_adapter.py, if one is installed
        from oracle import embree_adapter
        if embree_adapter.available():
            cpu_baseline["embree"] = embree_adapter.time_batch(sc, bounce, threads)
    except Exception as e:   # noqa: BLE001
        cpu_baseline["embree"] = {"error": str(e)[:200]}
    # The reference's OWN traversal kernel (oracle/_ref, built from RayAccelerator/Kernels.h with its own flags) on this same GPU and batch,
    # launched as the reference launches it (work-groups of 8, enqueue + clFinish).
    try:
        from oracle import ref_kernel
        if ref_kernel.built() and not S.xl_run:
            ref_res, ref_t = ref_kernel.run(blobs, bounce, sc["env"], repeats=5)
            hit = ref["triangle"] != 0xFFFFFFFF
            extras["reference_opencl_kernel_on_this_gpu"] = {
                "mrays_per_s": round(n / float(np.median(ref_t)) / 1e6, 1), "ms_per_launch": round(float(np.median(ref_t)) * 1e3, 3),
                "primId_agreement_with_engine": round(float((ref_res["triangle"][hit] == ref["triangle"][hit]).mean()), 6),
                "what": "Kernels.h `traversal`, -cl-fast-relaxed-math, local size 8, same 1M-ray diffuse batch, enqueue + clFinish"}
    except Exception as e:   # noqa: BLE001 - a missing OpenCL runtime must not fail the bench
        extras["reference_opencl_kernel_on_this_gpu"] = {"error": str(e)[:200]}
    return alg_bytes, alg_by_set, src, cpu_baseline, extras, visits


# ------------------------------------------------------------------------------------------------------------ the roofline object
def build_roofline(S, alg_bytes, alg_by_set, src, extras, xl):
    args = S.args
    prof = S.committed_profile()
    on_profiled_workload = bool(prof and S.full and args.mode == "weak" and S.standard_scene and S.host.quality == 1)
    pw = (prof or {}).get(args.workload, {}) if on_profiled_workload else {}
    pz = (prof or {}).get("diffuse_chained", {}) if (on_profiled_workload and args.workload == "diffuse") else {}
    traffic = pw.get("fabric_bytes_per_launch")
    step_s = S.elapsed / args.steps
    ceiling = S.gather_ceiling()
    alg_step = float(np.mean(alg_by_set)) if alg_by_set else alg_bytes      # the timed steps rotate through the sample sets
    iso_ms, iso_same_ms = S.iso_ms, S.iso_same_ms
    if not iso_ms:
        return None
    c = roofline_core(alg_step if iso_same_ms else alg_bytes, iso_ms, traffic, ceiling)
    if c is None:
        c = {"bound": "cu_gather_path" if ceiling else "hbm", "achieved": None, "peak": round(gather_peak_gbs(ceiling), 1) if ceiling else HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
             "traffic": traffic, "kernel_ms_avg": round(iso_ms, 4)}
    roofline = dict(c)
    steady = extras.get("batch_scaling", {}).get("steady_state_mrays_per_s")
    roofline.update({
        "kernel": S.kernel_name, "algorithmic_source": src,
        "bound_what": "the CU's vector-memory gather path: on this cache-resident scene (L2 hit rate %s) the algorithmic bytes are served by the L2s and the Infinity Cache and pass "
                      "through each CU's address / data-return path; `peak` = the rate at which this part gathers random 64-byte records at all, measured (tools/microbench/gather64.hip "
                      "mode 2: quad-cooperative LDS-DMA, 6 waves per SIMD, %s B/clk/CU x 256 CUs x 2.4 GHz; %s/gather64.txt).  `frac` is for the isolated launch `achieved` is defined on (a third "
                      "of which is ramp-up and drain); `frac_timed_region` / `frac_steady_state` hold the same bytes to the same roof in the chained loop and per marginal batch.  Within "
                      "that roof the step's dependent chain is what a wave waits on (`limiter.wave_time_split`); HBM is `hbm_algorithmic_frac` / `hbm_traffic_frac` and binds on "
                      "battlefield-synth-XL (`roofline_by_config`)" % (pw.get("l2_hit_rate", "n/a"), ceiling, S.profile_dir) if c.get("bound") == "cu_gather_path" else None,
        "frac_timed_region": round(alg_step / step_s / 1e9 / gather_peak_gbs(ceiling), 4) if (ceiling and alg_step) else None,
        "frac_steady_state": round(alg_bytes / ((1 << 20) / (steady * 1e6)) / 1e9 / gather_peak_gbs(ceiling), 4) if (ceiling and alg_bytes and steady) else None,
        "l2_aggregate_frac": round(alg_bytes / (iso_ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4) if alg_bytes else None,
        "kernel_ms_avg_same_batch": round(iso_same_ms, 4) if iso_same_ms else None,
        "kernel_ms_avg_note": "HIP events around the traversal kernel on the stream it is launched on, the kernel alone on the GPU, one launch at a time, BEFORE the warm-up steps "
                              "(`pre_timed_launches`): %d launches of one batch (mean of the last half = `kernel_ms_avg_same_batch`: that batch's rays are still in the Infinity Cache when "
                              "the next launch reads them), then as many rotating through the %d sample sets as the timed steps do (mean of the last half = `kernel_ms_avg`, what `achieved` is "
                              "computed from); rocprofv3 --kernel-trace of the same command with one lane and no chaining, same rotation: %s ms (%s/kernel_stats_one_lane.csv)" % (
                                  S.iso_n, len(S.d_sets), round(pw["kernel_ms_isolated"], 4) if pw.get("kernel_ms_isolated") else "n/a", S.profile_dir),
        "traffic_what": "L2-miss / fabric bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes, gfx950 correction of the guide), the kernel alone on the GPU as for "
                        "`achieved`; includes Infinity-Cache hits, so an upper bound of HBM traffic; %s/pmc_summary.json.  The x 2 is calibrated for this access pattern (%s/fetchcal.json)" % (S.profile_dir, S.profile_dir),
        # the timed region: launches are chained and overlap, so the rate is bytes per launch over the time the region spends per launch
        "timed_region": None if not alg_bytes else {
            "ms_per_step": round(step_s * 1e3, 4), "algorithmic_gbs": round(alg_step / step_s / 1e9, 1),
            "ray_batches_in_rotation": len(S.d_sets), "algorithmic_bytes_per_step_mean": int(alg_step),
            "kernel_event_ms_avg": round(float(np.mean(S.kernel_ms)), 4) if S.kernel_ms else None,
            "fabric_bytes_per_step_chained": pw.get("fabric_bytes_per_step_chained") or pz.get("fabric_bytes_per_launch"),
            "limiter_chained_instantiation": None if not pz else dict(
                {k: pz.get(k) for k in ("kernel", "wave_time_split", "td_busy_frac", "ta_busy_frac", "valu_busy_frac", "valu_lane_util", "salu_share", "l2_hit_rate",
                                        "vmem_rd_insts_per_ray", "valu_insts_per_ray")},
                what="the counters of the instantiation that is actually timed — the lazily chained one with in-kernel miss shading — from rocprofv3 --pmc passes over this "
                     "command with its default options (per step: summed over the chain's kernels / 20; rocprofv3 serialises dispatches under --pmc, so the chain's first "
                     "kernel does the work of all 20 steps alone)")},
        "limiter": None if not pw else {k: pw.get(k) for k in (
            "bound", "wave_time_split", "td_busy_frac", "ta_busy_frac", "valu_busy_frac", "valu_lane_util", "salu_share",
            "l2_hit_rate", "vmem_rd_insts_per_ray", "valu_insts_per_ray", "kernel_ms_isolated", "write_x_compulsory")},
        "limiter_steady_state": S.steady_state_profile() if (on_profiled_workload and args.workload == "diffuse") else None,
        "profile_source": (prof or {}).get("source"),
        "profile_stale": bool(prof["stale"]) if prof else None})
    # the other configs beside configs[2] (the headline): same definitions
    if on_profiled_workload and args.workload == "diffuse" and "coherent_1M" in extras:
        golden = S.golden
        pc = prof.get("coherent", {})
        keys = ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel_ms_avg", "hbm_algorithmic_frac", "hbm_traffic_frac")
        by = {"configs[2] 1M first-bounce diffuse": {k: roofline.get(k) for k in keys},
              "configs[1] 1M coherent primaries": roofline_core((golden.get("coherent_1M") or {}).get("bytes"), extras["coherent_1M"]["ms_per_step"], pc.get("fabric_bytes_per_launch"), ceiling)}
        for k, pk in (("configs[2] 1M first-bounce diffuse", pw), ("configs[1] 1M coherent primaries", pc)):
            if by[k] is not None:
                by[k]["limiter"] = {q: pk.get(q) for q in ("td_busy_frac", "valu_busy_frac", "valu_lane_util", "l2_hit_rate", "vmem_rd_insts_per_ray", "valu_insts_per_ray")} if pk else None
        pv = prof.get("v10_diffuse", {})
        if pv.get("kernel_ms_isolated") and alg_bytes:
            by["configs[2] with kernel_variant 50 (compressed 4-wide; the reference format's algorithmic bytes over ITS launch time)"] = dict(
                roofline_core(alg_bytes, pv["kernel_ms_isolated"], pv.get("fabric_bytes_per_launch"), ceiling) or {},
                limiter={q: pv.get(q) for q in ("wave_time_split", "td_busy_frac", "valu_busy_frac", "valu_lane_util", "l2_hit_rate", "vmem_rd_insts_per_ray", "valu_insts_per_ray")})
        if xl:
            by["battlefield-synth-XL 1M incoherent rays (HBM binds here)"] = xl.get("xl")
            by["battlefield-synth-XL 1M first-bounce diffuse (bench camera)"] = xl.get("xl_diffuse")
        extras["roofline_by_config"] = by
    if prof and prof["stale"]:
        print("bench: %s was taken with other kernel sources; re-run tools/profile_bench.sh + tools/summarize_profile.py" % S.profile_dir, file=sys.stderr)
    return roofline
