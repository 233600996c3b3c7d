#!/usr/bin/env python
"""Headline benchmark: Mrays/s of the intersect-batch hot path on MI355X.

A "step" is one pass of the path over one batch: 1,048,576 first-bounce diffuse
(incoherent) rays on the battlefield-synth stand-in scene per GPU (BASELINE.json
configs[2]; at N GPUs every rank traces its own 1M-ray batch = configs[3]'s 8M rays at
N=8, weak scaling, no data-path collective — rays never interact).  Inputs are resident
in HBM before the timed region.  The K steps are issued the way a caller of the C-ABI
issues batch after batch: racc_hip_intersect_device(lane = RACC_HIP_LANE_AUTO), i.e. the
engine rotates them over its lanes (≙ the reference's gpuSubmissionThreads queues,
RayAccelerator.cpp:711-717), and chains them: waves that run out of rays in one step's
batch go on with the next step's (include/racc_hip.h, chain_launches), so no step's drain
leaves the machine empty; every step has its own result array; the timed region ends when
every step's results are in HBM.  Output: ONE JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W]         # N > 1 as a plain command: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    ... bench.py --mode strong      # BASELINE configs[3] as written: ONE 8M-ray batch cut into N contiguous shards,
                                    # second figure with the RCCL all-gather of the hit records inside the timed region
"""
import argparse
import gc
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED_GBS = 6290.0      # same guide: float4-copy ceiling
RAYS_PER_BATCH = 1 << 20
NSETS = 8                      # ray batches that rotate through the steps of a timed region (sample sets of the same primary hits)
XL_RAY_SEED = 7
KERNEL_NAME = "traverseKernelV8"
L2_PEAK_GBS = 34500.0          # same guide, "L2 (per XCD)": 4 MiB x 8, ~34.5 TB/s aggregate
PCIE_GBS_PER_DIRECTION = 56.0  # page-locked copies on the GPU boxes, one direction alone (tools/microbench/pcie.hip; 49 + 49 with both at once)
# rocprofv3 summaries of THIS command (tools/profile_bench.sh <round>) + microbenchmark outputs: the latest round's that is committed
PROFILE_DIR = next(os.path.join("profiles", r) for r in ("r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", r, "derived.json")))
# what the committed counters depend on: the kernel, its launch policy (chunk, grid, chain), the device node order — and the tree builder
KERNEL_SOURCES = ("rayaccel_amd/csrc/racc_kernel_v8.inc", "rayaccel_amd/csrc/racc_kernel_v8_hot.inc", "rayaccel_amd/csrc/racc_device.inc", "rayaccel_amd/csrc/racc_launch.inc",
                  "rayaccel_amd/csrc/racc_scene_format.inc", "rayaccel_amd/csrc/racc_hip.hip", "rayaccel_amd/csrc/scene_build.cpp")
CU_CLOCK_HZ, CUS = 2.4e9, 256


def kernel_source_sha256():
    """Hash of the files the traversal kernel is compiled from; the committed profile records the one it was taken with."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def committed_profile():
    """Derived figures of the rocprofv3 passes committed under profiles/ (tools/summarize_profile.py), or None.
    Marked stale when the kernel sources changed since: the numbers then describe another kernel."""
    try:
        with open(os.path.join(ROOT, PROFILE_DIR, "derived.json")) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    d["stale"] = d.get("kernel_source_sha256") != kernel_source_sha256()
    return d


def steady_state_profile():
    """The limiter counters of 8M-ray launches (tools/r5_steady_pmc.sh -> profiles/<round>/steady_state_pmc.json): what the kernel does with its
    waves full, beside the isolated 1M-ray launch `limiter` describes.  None when missing; `stale` as for committed_profile()."""
    try:
        with open(os.path.join(ROOT, PROFILE_DIR, "steady_state_pmc.json")) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    keep = ("kernel_ms", "mrays_per_s", "wave_time_split", "valu_busy_frac", "valu_lane_util", "td_busy_frac", "ta_busy_frac", "salu_share",
            "valu_insts_per_ray", "vmem_rd_insts_per_ray", "l2_hit_rate")
    out = {"what": "the same counters for 8M-ray launches of the same kernel and tree (no ramp-up / drain share): %s/steady_state_pmc.json" % PROFILE_DIR}
    out.update({k: d.get(k) for k in keep})
    out["stale"] = d.get("kernel_source_sha256") != kernel_source_sha256()
    return out


def gather_ceiling():
    """Best rate at which a pure gather of random 64 B records runs on this part, bytes per clock per CU: measured by
    tools/microbench/gather64.hip (mode 2: quad-cooperative LDS-DMA), output committed under profiles/<round>/ by
    tools/microbench/run_microbench.sh.  None when that file is missing: no constant stands in for the measurement."""
    try:
        with open(os.path.join(ROOT, PROFILE_DIR, "microbench.json")) as f:
            return float(json.load(f)["gather64_ceiling_B_per_clk_per_CU"])
    except (OSError, ValueError, KeyError):
        return None


def roofline_core(alg, ms, traffic, ceiling):
    """The measurement contract's roofline fields for one (kernel, batch): algorithmic bytes per launch over the kernel's launch
    duration against the HBM peak; `traffic` = the L2-fabric bytes of the same launch from the committed rocprofv3 passes."""
    if not alg or not ms:
        return None
    return {"achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": int(alg), "kernel_ms_avg": round(ms, 4),
            "fabric_frac_of_hbm_peak": round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "fabric_frac_of_hbm_measured_ceiling": round(traffic / (ms * 1e-3) / 1e9 / HBM_MEASURED_GBS, 4) if traffic else None,
            "l1_gather_frac": round(alg / (ms * 1e-3) / (CUS * CU_CLOCK_HZ) / ceiling, 4) if ceiling else None}


def usable_cores():
    """Host cores this process may really use: CPU affinity capped by the cgroup CPU quota (the GPU boxes expose 256
    logical CPUs but grant 16 CPUs of quota; more threads than that only oversubscribes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=("weak", "strong"), default="weak",
                    help="weak (default): 1M rays per GPU per step; strong: one 8M-ray batch per step cut into N shards (configs[3])")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: also time the K steps with the RCCL all-gather of every step's hit records (racc_hip_allgather_results); implied by --mode strong")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (profiling runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra measurements (profiling runs)")
    ap.add_argument("--workload", choices=("diffuse", "coherent", "xl", "xl_diffuse"), default="diffuse",
                    help="diffuse (default, the headline): 1M first-bounce diffuse rays per step (configs[2]); coherent: the 1M primary rays (configs[1]); "
                         "xl / xl_diffuse: battlefield-synth-XL (25 M triangles, 1.3 GB on the device: past the Infinity Cache) with 1M incoherent rays / "
                         "its camera's 1M first-bounce diffuse rays — profiling runs of those configs")
    ap.add_argument("--grid", type=int, default=700, help="height-field resolution of battlefield-synth (700 = full)")
    ap.add_argument("--quality", type=int, default=1, choices=(0, 1, 2),
                    help="racc_host_build_options.quality of the scene build: 0 = the reference's builder (Bvh2.cpp restated, byte-identical to the oracle's), "
                         "1 (default) / 2 = the same reference-format blobs with one pair per leaf and re-inserted subtrees (fewer node visits per ray); "
                         "the line reports the quality-0 tree's figure beside it (`reference_builder_tree`)")
    ap.add_argument("--engine-opts", default="", help="JSON dict of racc_hip_options overrides (kernel A/B and profiling runs only)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # Started as a plain command (`python bench.py --gpus N ...`): become the launcher — one rank per GPU under
        # torch.distributed.run on a free local port, same arguments; rank 0's line is the only thing on stdout.
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if os.environ.get("RACC_BENCH_RANK_MARKERS"):      # (tests: evidence that this rank was started, written before anything can fail)
        open(os.path.join(os.environ["RACC_BENCH_RANK_MARKERS"], "rank%d_of_%d" % (rank, world)), "w").close()

    import numpy as np
    import torch                      # first: the engine then shares torch's HIP runtime in this process
    import torch.distributed as dist

    import rayaccel_amd as ra
    from rayaccel_amd import synth
    from rayaccel_amd.shard import shard_range

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU fallback)")
    # RACC_BENCH_BACKEND=gloo + RACC_BENCH_DEVICE=0 let the N>1 flow be rehearsed on a 1-GPU box (ranks share GPU 0);
    # the real thing is nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("RACC_BENCH_BACKEND", "nccl")
    device = int(os.environ.get("RACC_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- inputs (synthetic stand-in: the reference's battlefield.bin is unavailable) ----------
    full = args.grid == 700
    xl_run = args.workload in ("xl", "xl_diffuse")
    if xl_run:
        sc = synth.battlefield_synth_xl() if full else synth.battlefield_synth_xl(grid=args.grid)
    else:
        sc = synth.battlefield_synth() if full else synth.battlefield_synth(grid=args.grid, boxes=args.grid * 6, quads=args.grid * 28)
    # (N ranks share the host's cores while they set up: the BVH build of every rank takes its share, not all of them)
    os.environ.setdefault("RACC_BUILD_THREADS", str(max(1, usable_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))))
    host = ra.HostScene(sc["vertices"], sc["indices"], quality=args.quality)
    engine_opts = dict()      # (--engine-opts '{"time_kernels":1}' adds an event pair around every traversal kernel: `timed_region.kernel_event_ms_avg`; it costs ~1 % of `value`)
    engine_opts.update(json.loads(args.engine_opts) if args.engine_opts else {})
    ctx = ra.Context(device=device, **engine_opts)
    lanes = ctx.lanes
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])

    primary, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    primary_hits = ctx.intersect(scene, env, primary)                       # GPU path, host buffers
    ray_sets = None
    if args.workload == "coherent":
        bounce = primary                       # configs[1]: the timed batch is the coherent primary batch itself
        total_rays = world * len(bounce)
    elif args.workload == "xl":
        # incoherent: origins and directions uniform over the scene; NSETS different batches rotate through the steps
        ray_sets = [synth.random_rays(RAYS_PER_BATCH, XL_RAY_SEED + rank + 1000 * k) for k in range(NSETS)]
        bounce = ray_sets[0]
        total_rays = world * len(bounce)
    elif args.mode == "weak":
        # NSETS sample sets of the same primary hits rotate through the steps: no step re-traces the rays of the step before it
        # (a caller never issues the same rays twice); rank r starts the rotation at set r
        ray_sets = synth.diffuse_bounce_batches(sc, primary, primary_hits, RAYS_PER_BATCH, [(rank + k) % NSETS for k in range(NSETS)])
        bounce = ray_sets[0]
        total_rays = world * len(bounce)
    else:       # strong: configs[3], 8 sample sets = one 8M-ray batch; this rank's contiguous shard of it
        whole = np.concatenate([synth.diffuse_bounce_rays(sc, primary, primary_hits, RAYS_PER_BATCH, first_sample=k) for k in range(8)])
        total_rays = len(whole)
        b, e = shard_range(total_rays, rank, world)
        bounce = np.ascontiguousarray(whole[b:e])
        del whole
    n = len(bounce)
    if ray_sets is None:
        ray_sets = [bounce]      # configs[1]: a fixed camera's primaries ARE the same rays every frame; strong mode: one 8M-ray batch

    d_sets = [torch.from_numpy(r.view(np.float32).reshape(n, 8).copy()).cuda() for r in ray_sets]
    d_rays = d_sets[0]
    # One result array per batch issued between two waits: chained launches (racc_hip_options::chain_launches, the default) keep a
    # batch's arrays until the wait returns.  16 MiB each: 3.1 GiB for the default 200 steps.  The ray array is read-only and shared.
    outs = [torch.zeros((n, 4), dtype=torch.float32, device="cuda") for _ in range(min(max(args.steps, args.warmup, lanes, 2), 1024))]      # (beyond 1024 steps arrays repeat: every step writes the same bits)
    d_out = outs[0]
    torch.cuda.synchronize()

    def run_overlapped(steps):
        """`steps` batches, issued like a caller of the C-ABI issues them: the engine rotates the lanes."""
        for k in range(steps):
            ctx.intersect_device(scene, env, d_sets[k % len(d_sets)].data_ptr(), outs[k % len(outs)].data_ptr(), n, lane=ra.LANE_AUTO)
        ctx.wait(ra.LANE_AUTO)

    def drain_kernel_times():
        if not engine_opts.get("time_kernels"):
            return []
        return [t for lane in range(lanes) for t in ctx.kernel_times(lane)]

    # The traversal kernel alone on the GPU, one launch at a time (HIP events around the kernel on the stream it is launched on):
    # the launch duration `roofline.achieved` is computed from — in the timed region launches are chained, a kernel there either
    # works through many batches or finds nothing left, so no per-launch duration exists.  Untimed, before the warm-up.  The
    # duration settles only after the GPU has been busy for ~15 ms (10 launches: 0.386 ms, 100: 0.373; rocprofv3's one-lane
    # trace of the committed profile: 0.371), so 60 launches are timed and the mean of the last 30 is reported.
    # (every rank does it: at N > 1 all GPUs enter the timed region in the same state, and the line reports rank 0's)
    # These launches run BEFORE the warm-up steps and are reported as `pre_timed_launches`: the GPU's clocks are up when the timed
    # region starts (RACC_BENCH_ISO_LAUNCHES=0 skips them — the profiling passes do, so that their last K traversal dispatches are the K steps).
    # No collector pause inside the K steps or between the warm-up and them: a full collection of this process takes ~35 ms, during
    # which the GPU would sit idle and drop its clocks (seen in a rocprofv3 timeline of this command).  Collected here, once, before the
    # isolated launches; switched back on after the timed region.
    gc.collect()
    gc.disable()
    iso_n = int(os.environ.get("RACC_BENCH_ISO_LAUNCHES", "60"))
    iso_ms = None
    iso_same_ms = None
    if iso_n > 0:
        iso_all = ctx.intersect_device_timed(scene, env, d_rays.data_ptr(), outs[-1].data_ptr(), n, iso_n)
        iso_ms = float(np.mean(iso_all[len(iso_all) // 2:]))
        if len(d_sets) > 1:
            # ... and the same number of launches ROTATING through the sample sets, as the timed steps and the committed rocprofv3 passes do
            # (the block above re-traces ONE batch, whose 32 MiB of rays are still in the Infinity Cache when the next launch reads them):
            # this is the duration `roofline.achieved` is computed from; the one-batch figure stays beside it (`kernel_ms_avg_same_batch`).
            # One launch per call (the timed entry takes one ray array): the stream drains between launches, the clocks are up from the block above.
            iso_same_ms = iso_ms
            rot = [ctx.intersect_device_timed(scene, env, d_sets[i % len(d_sets)].data_ptr(), outs[-1].data_ptr(), n, 1)[0] for i in range(iso_n)]
            iso_ms = float(np.mean(rot[len(rot) // 2:]))

    if args.warmup:
        run_overlapped(args.warmup)
    drain_kernel_times()

    # ---- timed region: exactly K steps, barrier + device sync on both sides --------------------
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_overlapped(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    gc.enable()
    kernel_ms = drain_kernel_times()            # HIP events around each traversal kernel, on the stream it ran on
    for k in range(len(d_sets), min(args.steps, len(outs))):      # step k traced sample set k mod NSETS: same rays, same bits (the sets themselves are held to the oracle below)
        if not torch.equal(outs[k].view(torch.int32), outs[k % len(d_sets)].view(torch.int32)):
            sys.exit("bench: step %d of the timed region produced other results than step %d (same rays)" % (k, k % len(d_sets)))
    per_rank, comm_ranks = [elapsed], 1
    if world > 1:
        # SCALE-day hardening: rank 0's line must not depend on another rank's teardown.  If the gather of the elapsed times has not come back
        # after RACC_BENCH_GATHER_TIMEOUT seconds (default 120; a rank died, the fabric hangs), rank 0 prints a line from ITS OWN elapsed time,
        # marked "partial": true, and the process exits — instead of hanging until the driver's limit with nothing on stdout.
        import threading
        gather_timeout = float(os.environ.get("RACC_BENCH_GATHER_TIMEOUT", "120"))

        def give_up():
            if rank == 0:
                print(json.dumps({"metric": "Mrays/s", "value": round(total_rays * args.steps / elapsed / 1e6, 1), "unit": "Mrays/s", "n_gpus": world,
                                  "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                                  "scaling": args.mode, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "partial": True,
                                  "partial_reason": "the all-gather of the ranks' elapsed times did not return within %.0f s: value = all ranks' rays over RANK 0's own time, "
                                                    "not the maximum over ranks" % gather_timeout,
                                  "config": {"workload": "battlefield-synth (stand-in), 1M 1st-bounce diffuse rays per GPU per step", "rays_per_gpu": n}}), flush=True)
            os._exit(3)
        watchdog = threading.Timer(gather_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        cdev = "cuda" if backend == "nccl" else "cpu"
        mine = torch.tensor([elapsed, 1.0], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        if os.environ.get("RACC_BENCH_TEST_HANG") == str(rank):      # (tests: this rank never joins the gather)
            time.sleep(3600)
        dist.all_gather(every, mine)                                   # RCCL over xGMI when backend == nccl
        if cdev == "cuda":
            torch.cuda.synchronize()
        watchdog.cancel()
        per_rank = [float(t[0].item()) for t in every]
        comm_ranks = int(round(sum(float(t[1].item()) for t in every)))      # ranks that took part in the collective
        elapsed = max(per_rank)

    value = total_rays * args.steps / elapsed / 1e6
    launch = ctx.launch_info()
    d_ref_bits = d_out.view(torch.int32).clone()      # the default kernel's records of the timed batch (the extras reuse the result arrays)
    set_bits = [outs[k].view(torch.int32).clone() for k in range(1, min(len(d_sets), args.steps, len(outs)))]      # ... and of the other sample sets

    # ---- optional extras, all outside the timed region ------------------------------------------
    extras = {}
    if world > 1 and backend == "nccl" and (args.gather or args.mode == "strong"):
        # (opt-in: the default line must not depend on a second collective library instance coming up on every rank)
        # RCCL all-gather of the Result shards over xGMI through the C-ABI's own entry (racc_hip_allgather_results binds
        # librccl; only a GPU-side consumer that needs every hit on every GPU needs it).  Second timed figure: the same K
        # steps with the gather of every step's results inside the region.
        uid = [ra.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = ra.Comm(ctx, uid[0], rank, world)
        per = -(-total_rays // world) if args.mode == "strong" else n
        gathered = torch.empty((world * per, 4), dtype=torch.float32, device="cuda")
        send = torch.zeros((per, 4), dtype=torch.float32, device="cuda")
        comm.allgather_results(send.data_ptr(), gathered.data_ptr(), per)
        torch.cuda.synchronize(); barrier()
        t1 = time.perf_counter()
        for k in range(args.steps):
            lane = k % lanes
            ctx.intersect_device(scene, env, d_rays.data_ptr(), send.data_ptr(), n, lane=lane)
            ctx.wait(lane)
            comm.allgather_results(send.data_ptr(), gathered.data_ptr(), per)
        ctx.synchronize(); torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        extras["with_allgather_of_results"] = {"mrays_per_s": round(total_rays * args.steps / float(dt.item()) / 1e6, 1),
                                               "ms_per_step": round(float(dt.item()) / args.steps * 1e3, 4),
                                               "bytes_gathered_per_step": int(gathered.numel() * 4),
                                               "how": "racc_hip_allgather_results (C-ABI -> ncclAllGather), one message per rank, trace and gather serialised per step"}
        comm.destroy()
    # (N > 1: no untimed extras — rank 0 would still be measuring while the other ranks tear the process group down)
    if rank == 0 and not args.no_extras and world == 1:
        def timed_serial(rays_t, out_t, iters):
            return ctx.intersect_device_timed(scene, env, rays_t.data_ptr(), out_t.data_ptr(), rays_t.shape[0], iters)
        # one launch at a time (what round 1 reported as `value`): the launch's drain is exposed
        timed_serial(d_rays, outs[1], 2)
        t1 = time.perf_counter()
        serial_ms = timed_serial(d_rays, outs[1], args.steps)
        extras["one_launch_at_a_time"] = {"mrays_per_s": round(args.steps * n / (time.perf_counter() - t1) / 1e6, 1),
                                          "kernel_ms_avg": round(float(np.mean(serial_ms)), 4)}
        if not torch.equal(outs[1].view(torch.int32), d_ref_bits):   # bit compare (a miss id reads as NaN in f32)
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
        ctx.intersect(scene, env, bounce, res_host)
        t1 = time.perf_counter()
        for _ in range(3):
            ctx.intersect(scene, env, bounce, res_host)
        extras["host_buffers_pcie_inclusive_mrays_per_s"] = round(3 * n / (time.perf_counter() - t1) / 1e6, 1)
        if not np.array_equal(res_host.view(np.uint32).reshape(-1, 4), d_ref_bits.cpu().numpy().view(np.uint32)):
            sys.exit("bench: the host-buffer path changed the results")
        # ... and page-locked arrays (racc_hip_register_host, what racc::createContext does with its stream block) the way a host
        # application binds the C-ABI: tools/host_path_bench.py in a process of its own, without torch.  (torch ships its own, older HIP
        # runtime; the engine shares it in THIS process, and the same pipeline then moves 40-47 instead of 54 GB/s into the GPU: measured
        # both ways, tools/gpu_hostpipe.py.)  One batch at a time through the blocking entry — cut into slices so that copies run beside
        # kernels — and batches issued back to back with racc_hip_intersect_async on rotating lanes: copy-in, kernels and copy-out of
        # consecutive batches in flight (racc_hostpath.inc).  PCIe is full duplex: the directions are reported separately, each against
        # what one direction delivers alone on this box.
        if world == 1:
            import subprocess
            try:
                p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_path_bench.py"), "--grid", str(args.grid), "--device", str(device),
                                    "--link-gbs", str(PCIE_GBS_PER_DIRECTION)], capture_output=True, text=True, timeout=600, cwd=ROOT)
                extras["host_buffers_page_locked"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else {"error": (p.stderr or p.stdout)[-300:]}
            except Exception as e:   # noqa: BLE001
                extras["host_buffers_page_locked"] = {"error": str(e)[:200]}

        # The quality-0 tree (the reference's builder, byte-identical to the oracle's restatement of Bvh2.cpp) in the SAME loop as `value`,
        # same context, same rays, and one launch at a time: what the tree post-processing of racc_host_scene_build_ex buys.
        if args.quality and world == 1 and args.mode == "weak" and not xl_run:
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
            same_prim = int((outs[0].view(torch.int32)[:, 0] == d_ref_bits[:, 0]).sum().item())
            extras["reference_builder_tree"] = {
                "mrays_per_s_same_loop_as_value": round(n * args.steps / dt0 / 1e6, 1), "ms_per_step": round(dt0 / args.steps * 1e3, 4),
                "kernel_ms_avg_one_launch_at_a_time": round(float(np.mean(iso0[len(iso0) // 2:])), 4),
                "inner_nodes": len(h0.nodes), "pairs": int(h0.pair_count),
                "primIds_equal_to_the_quality_tree": "%d of %d" % (same_prim, n),
                "what": "racc_host_build_options.quality = 0: Bvh2.cpp:257-535 restated, byte-identical to the oracle's builder; t/u/v of the two trees agree to "
                        "rounding, primIds up to exact-distance ties (tests/test_quality_build.py, tests/test_gpu_quality.py)"}
            scene0.destroy()
            del h0

        # Batch-size scaling of the traversal kernel (same diffuse rays, 8 sample sets): T(N) = fixed + per-ray cost.
        if world == 1 and full and args.mode == "weak":
            many = np.concatenate(ray_sets) if len(ray_sets) == 8 else np.concatenate(synth.diffuse_bounce_batches(sc, primary, primary_hits, RAYS_PER_BATCH, range(8)))
            d_many = torch.from_numpy(many.view(np.float32).reshape(len(many), 8).copy()).cuda()
            d_many_out = torch.zeros((len(many), 4), dtype=torch.float32, device="cuda")
            scaling = {}
            for nn in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 23):
                ctx.intersect_device_timed(scene, env, d_many.data_ptr(), d_many_out.data_ptr(), nn, 2)
                scaling[str(nn)] = round(float(np.median(ctx.intersect_device_timed(scene, env, d_many.data_ptr(), d_many_out.data_ptr(), nn, 7))), 4)
            slope = (scaling[str(1 << 23)] - scaling[str(1 << 20)]) / 7.0          # ms per 2^20 rays
            extras["batch_scaling"] = {"kernel_ms_by_rays": scaling, "steady_state_mrays_per_s": round((1 << 20) / slope / 1e3, 1),
                                       "fixed_ms": round(scaling[str(1 << 20)] - slope, 4)}
            # the compressed 4-wide kernel (kernel_variant 50, racc_kernel_v10.inc) on the same batches, its own
            # context, same scene blobs: single launches, and the SAME loop as the timed region (K chained steps after W warm-up)
            try:
                with ra.Context(device=device, kernel_variant=50, time_kernels=0) as wctx:
                    wscene = wctx.upload_scene(host.nodes, host.pairs, host.remap)
                    wenv = wctx.create_environment(sc["env"])
                    wide = {}
                    for nn in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 23):
                        wctx.intersect_device_timed(wscene, wenv, d_many.data_ptr(), d_many_out.data_ptr(), nn, 2)
                        wide[str(nn)] = round(float(np.median(wctx.intersect_device_timed(wscene, wenv, d_many.data_ptr(), d_many_out.data_ptr(), nn, 7))), 4)
                    wslope = (wide[str(1 << 23)] - wide[str(1 << 20)]) / 7.0

                    def wrun(steps):
                        for k in range(steps):
                            wctx.intersect_device(wscene, wenv, d_rays.data_ptr(), outs[k % len(outs)].data_ptr(), n, lane=ra.LANE_AUTO)
                        wctx.wait(ra.LANE_AUTO)
                    if args.warmup:
                        wrun(args.warmup)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    wrun(args.steps)
                    torch.cuda.synchronize()
                    wdt = time.perf_counter() - t1
                    cmp = outs[min(args.steps, len(outs)) - 1]
                    differing = int((cmp.view(torch.int32) != d_ref_bits).any(dim=1).sum().item())
                    extras["compressed_wide_kernel_variant_50"] = {
                        "mrays_per_s_same_loop_as_value": round(n * args.steps / wdt / 1e6, 1), "ms_per_step": round(wdt / args.steps * 1e3, 4),
                        "kernel_ms_by_rays": wide, "default_kernel_ms_by_rays": {k: scaling[k] for k in wide},
                        "steady_state_mrays_per_s": round((1 << 20) / wslope / 1e3, 1), "fixed_ms": round(wide[str(1 << 20)] - wslope, 4),
                        "records_differing_from_default_in_1M": differing,
                        "note": "64 B 4-wide nodes, child boxes quantised conservatively to 8 bits: half the node bytes and vector-memory instructions per ray; "
                                "same closest hit as the default kernel except exact-distance ties and arbiter-confirmed closer hits (DESIGN.md §3, §5)"}
                    wscene.destroy(); wenv.destroy()
            except ra.RaccError as e:
                extras["compressed_wide_kernel_variant_50"] = {"error": str(e)}
            del d_many, d_many_out

        # BASELINE configs[4]: the path tracer, 1920x1080, end to end on this GPU.  Device-resident consumer (generation and
        # shading kernels around racc_hip_intersect_device) at 64 spp; the reference-shaped consumer (spawn/shade callbacks on
        # host threads through racc::render, PCIe both ways) at 8 spp.  Both render the same image (tests/test_gpu_pathtracer.py).
        if world == 1 and full and args.mode == "weak":
            import tempfile
            from rayaccel_amd.engine import path_trace
            tmp = tempfile.NamedTemporaryFile(suffix=".bin", delete=False)
            tmp.close()
            try:
                synth.write_scene_bin(tmp.name, sc, viewport=(1920, 1080))
                os.environ["RACC_BUILD_QUALITY"] = str(args.quality)      # the consumers build their own scene through racc_host_scene_build: same tree as the headline
                _, sg = path_trace(tmp.name, 1920, 1080, 0, 64, device=device, shading="gpu")
                _, sh = path_trace(tmp.name, 1920, 1080, 0, 8, device=device, shading="cpu", cpu_threads=usable_cores())
                # racc::render with callbacks that cost nothing (spawn = memcpy of a pre-generated 128x128 tile, shade empty): what the ray-stream
                # state machine + the host RayStream path sustain by themselves — the ceiling of the drop-in API (tests/cpp/render_check.cpp)
                sched = None
                try:
                    import subprocess
                    pr = subprocess.run([os.path.join(ROOT, "tests", "cpp", "render_check"), tmp.name, "--null-callbacks", "1920", "1080", "16", "4"],
                                        capture_output=True, text=True, timeout=300, env=dict(os.environ, RACC_CPU_THREADS=str(usable_cores())))
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

    # ---- battlefield-synth-XL: the regime in which HBM can bind (rank 0, N = 1) --------------------
    xl = None
    if rank == 0 and world == 1 and full and args.mode == "weak" and args.workload == "diffuse" and not args.no_extras:
        xl = {}
        prof_x = committed_profile() or {}
        sx = synth.battlefield_synth_xl()
        hx = ra.HostScene(sx["vertices"], sx["indices"], quality=args.quality)
        scene_x = ctx.upload_scene(hx.nodes, hx.pairs, hx.remap)
        env_x = ctx.create_environment(sx["env"])
        hits_x = ctx.intersect(scene_x, env_x, primary)
        xl_batches = (("xl", "1M incoherent rays (origins and directions uniform over the scene)", synth.random_rays(RAYS_PER_BATCH, XL_RAY_SEED)),
                      ("xl_diffuse", "1M first-bounce diffuse rays of the bench camera", synth.diffuse_bounce_rays(sx, primary, hits_x, RAYS_PER_BATCH)))
        for key, what, rays_x in xl_batches:
            d_rx = torch.from_numpy(rays_x.view(np.float32).reshape(len(rays_x), 8).copy()).cuda()
            d_ox = torch.zeros((len(rays_x), 4), dtype=torch.float32, device="cuda")
            ms_x = ctx.intersect_device_timed(scene_x, env_x, d_rx.data_ptr(), d_ox.data_ptr(), len(rays_x), 40)
            ms_x = float(np.mean(ms_x[len(ms_x) // 2:]))
            alg_x, src_x = None, None
            if not args.no_cpu_baseline:
                from oracle import oracle            # checker only
                ref_x, nv_x, np_x, _ = oracle.traverse(hx.blobs(), rays_x, env=sx["env"], counters=True, threads=usable_cores())
                alg_x, src_x = oracle.algorithmic_bytes(ref_x, nv_x, np_x), "oracle counters, live"
                got_x = d_ox.cpu().numpy().view(ra.RESULT_DTYPE).reshape(-1)
                hit_x = ref_x["triangle"] != 0xFFFFFFFF
                if not np.array_equal(got_x["triangle"], ref_x["triangle"]) or any(
                        not np.array_equal(got_x[f][hit_x].view(np.uint32), ref_x[f][hit_x].view(np.uint32)) for f in ("t", "u", "v")):
                    sys.exit("bench: GPU results on battlefield-synth-XL (%s) differ from the oracle — refusing to report a number" % key)
            else:
                try:
                    with open(os.path.join(ROOT, "tests", "golden", "algorithmic_bytes.json")) as f:
                        gx = json.load(f)
                        alg_x, src_x = (gx["quality%d" % args.quality] if args.quality else gx)[key + "_1M"]["bytes"], "tests/golden/algorithmic_bytes.json"
                except (OSError, KeyError, ValueError):
                    pass
            px = prof_x.get(key, {})
            r = roofline_core(alg_x, ms_x, px.get("fabric_bytes_per_launch"), gather_ceiling()) or {"kernel_ms_avg": round(ms_x, 4)}
            r.update({"workload": "battlefield-synth-XL, %d triangles, %s" % (len(sx["indices"]), what), "mrays_per_s": round(len(rays_x) / ms_x / 1e3, 1),
                      "algorithmic_source": src_x, "device_bytes": int(scene_x.info["device_bytes"]),
                      "traffic_frac_of_algorithmic": round(px["fabric_bytes_per_launch"] / alg_x, 4) if (px.get("fabric_bytes_per_launch") and alg_x) else None,
                      "limiter": {q: px.get(q) for q in ("td_busy_frac", "ta_busy_frac", "valu_busy_frac", "l2_hit_rate", "kernel_ms_isolated", "fetch_bytes_per_launch", "write_bytes_per_launch")} if px else None})
            xl[key] = r
            del d_rx, d_ox
        scene_x.destroy(); env_x.destroy()
        del hx, sx

    # ---- roofline + CPU baseline (rank 0) --------------------------------------------------------
    roofline, cpu_baseline = None, None
    if rank == 0:
        avg_kernel_ms = float(np.mean(kernel_ms)) if kernel_ms else None
        alg_bytes, src = None, None
        golden = {}
        try:
            with open(os.path.join(ROOT, "tests", "golden", "algorithmic_bytes.json")) as f:
                golden = json.load(f)
        except (OSError, ValueError):
            pass
        golden_key = {"diffuse": "diffuse_1M_sample0", "coherent": "coherent_1M", "xl": "xl_1M", "xl_diffuse": "xl_diffuse_1M"}[args.workload]
        if args.quality:
            golden = golden.get("quality%d" % args.quality, {})
        alg_by_set = None
        if not args.no_cpu_baseline and world == 1 and args.mode == "weak":       # the CPU legs run at N=1 only
            from oracle import oracle            # checker / CPU leg only; never on the product path
            blobs = host.blobs()
            ref, nv, npairs, _ = oracle.traverse(blobs, bounce, env=sc["env"], counters=True, threads=usable_cores())
            alg_bytes, src = oracle.algorithmic_bytes(ref, nv, npairs), "oracle counters, live, on the blobs the GPU traverses (racc_host_build_options.quality = %d)" % args.quality
            alg_by_set = [alg_bytes]
            for k, bits in enumerate(set_bits, 1):      # the other sample sets of the rotation: every record against the oracle as well
                ref_k, nv_k, np_k, _ = oracle.traverse(blobs, ray_sets[k], env=sc["env"], counters=True, threads=usable_cores())
                got_k = bits.cpu().numpy().view(ra.RESULT_DTYPE).reshape(-1)
                hit_k = ref_k["triangle"] != 0xFFFFFFFF
                if not np.array_equal(got_k["triangle"], ref_k["triangle"]) or any(
                        not np.array_equal(got_k[f][hit_k].view(np.uint32), ref_k[f][hit_k].view(np.uint32)) for f in ("t", "u", "v")):
                    sys.exit("bench: GPU results of sample set %d differ from the oracle — refusing to report a number" % k)
                alg_by_set.append(oracle.algorithmic_bytes(ref_k, nv_k, np_k))
            got = d_ref_bits.cpu().numpy().view(ra.RESULT_DTYPE).reshape(-1)
            hit = ref["triangle"] != 0xFFFFFFFF
            if not np.array_equal(got["triangle"], ref["triangle"]) or any(
                    not np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)) for f in ("t", "u", "v")) or any(
                    not np.allclose(got[f][~hit], ref[f][~hit], rtol=1e-5, atol=1e-5) for f in ("t", "u", "v")):
                sys.exit("bench: GPU results (primId, t, u, v bits; miss colours to 1e-5) differ from the oracle — refusing to report a number")
            threads = usable_cores()
            cpu_out = np.zeros(n, oracle.RESULT_DTYPE)
            oracle.traverse(blobs, bounce, env=sc["env"], threads=threads, out=cpu_out)          # warm-up: faults pages, starts clocks
            t1 = time.perf_counter()
            oracle.traverse(blobs, bounce, env=sc["env"], threads=threads, out=cpu_out)
            one = time.perf_counter() - t1
            repeat = int(min(64, max(2, 12.0 / max(one, 1e-3))))                                 # ~10-15 s of CPU work in total
            times = []
            for _ in range(3):
                t1 = time.perf_counter()
                oracle.traverse(blobs, bounce, env=sc["env"], threads=threads, repeat=repeat, out=cpu_out)
                times.append((time.perf_counter() - t1) / repeat)
            cpu_baseline = {"value": round(n / float(np.median(times)) / 1e6, 2), "unit": "Mrays/s", "cores": threads,
                            "kind": "port",
                            "sample": "the full 1,048,576-ray batch of the timed workload, %d passes per timing x 3 timings (median), %d pthreads x "
                                      "1024-ray slices; SCALAR BVH2 port of the reference's traversal, not Embree-class "
                                      "(the reference's CPU path is binary-only Embree 2.x bvh8/AVX2, unavailable here; "
                                      "oracle/embree_adapter.py adds a row when a system Embree exists)" % (repeat, threads)}
            try:        # optional second CPU row: a system Embree through oracle/embree_adapter.py, if one is installed
                from oracle import embree_adapter
                if embree_adapter.available():
                    cpu_baseline["embree"] = embree_adapter.time_batch(sc, bounce, threads)
            except Exception as e:   # noqa: BLE001
                cpu_baseline["embree"] = {"error": str(e)[:200]}
            # The reference's OWN traversal kernel (oracle/_ref, built from RayAccelerator/Kernels.h with its own flags) on this
            # same GPU and batch, launched as the reference launches it (work-groups of 8, enqueue + clFinish).
            try:
                from oracle import ref_kernel
                if ref_kernel.built() and not xl_run:
                    ref_res, ref_t = ref_kernel.run(blobs, bounce, sc["env"], repeats=5)
                    hit = ref["triangle"] != 0xFFFFFFFF
                    agree = float((ref_res["triangle"][hit] == ref["triangle"][hit]).mean())
                    extras["reference_opencl_kernel_on_this_gpu"] = {
                        "mrays_per_s": round(n / float(np.median(ref_t)) / 1e6, 1), "ms_per_launch": round(float(np.median(ref_t)) * 1e3, 3),
                        "primId_agreement_with_engine": round(agree, 6),
                        "what": "Kernels.h `traversal`, -cl-fast-relaxed-math, local size 8, same 1M-ray diffuse batch, enqueue + clFinish"}
            except Exception as e:   # noqa: BLE001 - a missing OpenCL runtime must not fail the bench
                extras["reference_opencl_kernel_on_this_gpu"] = {"error": str(e)[:200]}
        elif full and args.mode == "weak":
            alg_bytes, src = (golden.get(golden_key) or {}).get("bytes"), "tests/golden/algorithmic_bytes.json"
        prof = committed_profile()
        on_profiled_workload = bool(prof and full and args.mode == "weak")
        pw = (prof or {}).get(args.workload, {}) if on_profiled_workload else {}
        traffic = pw.get("fabric_bytes_per_launch")
        step_s = elapsed / args.steps
        ceiling = gather_ceiling()
        alg_step = float(np.mean(alg_by_set)) if alg_by_set else alg_bytes      # the timed steps rotate through the sample sets
        if iso_ms:
            # The contract's roofline: `achieved` = the ALGORITHMIC bytes of SURVEY §8(d) per launch (every node / pair the reference's
            # traversal order touches, counted by the oracle) over the traversal kernel's launch duration (HIP events, the kernel alone
            # on the GPU), against the 8 TB/s of HBM.  On battlefield-synth the fraction comes out ABOVE 1: the 55 MB scene lives in the
            # L2s and the Infinity Cache, which serve nearly all of those bytes — the work is done (every record of the timed batch is
            # compared with the oracle above), HBM is simply not what bounds this kernel there: `bound_actual` holds the same bytes to
            # the two levels they do pass through (the L2s' aggregate bandwidth; the CU's vector-memory return path, measured by a
            # committed microbenchmark), `traffic` = what did cross the L2-fabric boundary per launch (rocprofv3 FETCH_SIZE x2 +
            # WRITE_SIZE, same isolated mode, committed profile), `limiter` = the counters that say what binds it.  Where HBM CAN bind
            # is `roofline_by_config["battlefield-synth-XL ..."]`: 1.3 GB of scene, incoherent rays, frac < 1, traffic > the algorithmic bytes.
            # (the isolated launches rotate through the sample sets: the bytes of a launch are the sets' mean, within 0.1 % of any one set's)
            c = roofline_core(alg_step if iso_same_ms else alg_bytes, iso_ms, traffic, ceiling) or {"achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": traffic, "kernel_ms_avg": round(iso_ms, 4)}
            roofline = {"bound": "hbm"}
            roofline.update(c)
            roofline.update({
                "kernel": KERNEL_NAME, "algorithmic_source": src,
                "kernel_ms_avg_same_batch": round(iso_same_ms, 4) if iso_same_ms else None,
                "kernel_ms_avg_note": "HIP events around the traversal kernel on the stream it is launched on, the kernel alone on the GPU, one launch at a time, BEFORE the warm-up steps "
                                      "(`pre_timed_launches`): %d launches of one batch (mean of the last %d = `kernel_ms_avg_same_batch`: that batch's rays are still in the Infinity Cache "
                                      "when the next launch reads them), then %d launches rotating through the %d sample sets as the timed steps do (mean of the last %d = `kernel_ms_avg`, "
                                      "what `achieved` is computed from); rocprofv3 --kernel-trace of the same command with one lane and no chaining, same rotation: %s ms (%s/kernel_stats_one_lane.csv)" % (
                                          iso_n, iso_n - iso_n // 2, iso_n if iso_same_ms else 0, len(d_sets), iso_n - iso_n // 2,
                                          round(pw["kernel_ms_isolated"], 4) if pw.get("kernel_ms_isolated") else "n/a", PROFILE_DIR),
                "frac_is": "algorithmic bytes / kernel duration / HBM peak — NOT a utilisation of HBM when it exceeds `traffic`'s share: see bound_actual",
                "bound_actual": None if not alg_bytes else {
                    "what": "the algorithmic bytes against the levels they pass through on this scene (cache-resident: L2 hit rate %s)" % pw.get("l2_hit_rate", "n/a"),
                    "l2_aggregate": {"peak_gbs": L2_PEAK_GBS, "frac_isolated_launch": round(alg_bytes / (iso_ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4),
                                     "frac_timed_region": round(alg_step / step_s / 1e9 / L2_PEAK_GBS, 4)},
                    "cu_gather_path": None if not ceiling else {
                        "measured_ceiling_B_per_clk_per_CU": ceiling,
                        "frac_isolated_launch": round(alg_bytes / (iso_ms * 1e-3) / (CUS * CU_CLOCK_HZ) / ceiling, 4),
                        "frac_timed_region": round(alg_step / step_s / (CUS * CU_CLOCK_HZ) / ceiling, 4),
                        "frac_steady_state": (round(alg_bytes / ((1 << 20) / (extras["batch_scaling"]["steady_state_mrays_per_s"] * 1e6)) / (CUS * CU_CLOCK_HZ) / ceiling, 4)
                                              if "batch_scaling" in extras else None),
                        "note": "256 CUs x 2.4 GHz; ceiling = tools/microbench/gather64.hip mode 2 (64 random 64 B records per wave through quad-cooperative LDS-DMA, "
                                "6 waves/SIMD), output in %s/gather64.txt" % PROFILE_DIR},
                    "hbm": {"peak_gbs": HBM_PEAK_GBS, "fabric_traffic_frac_of_peak": c.get("fabric_frac_of_hbm_peak"),
                            "traffic_frac_of_algorithmic": round(traffic / alg_bytes, 4) if traffic else None}},
                "traffic_what": "L2-miss / fabric bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes, gfx950 correction of the guide), "
                                "the kernel alone on the GPU as for `achieved`; includes Infinity-Cache hits, so an upper bound of HBM traffic; %s/pmc_summary.json.  The x 2 is calibrated for "
                                "this access pattern: a gather of 64-byte records through the kernel's own fetch moves whole 128-byte lines, each tallied at 64 B (%s/fetchcal.json)" % (PROFILE_DIR, PROFILE_DIR),
                # the timed region: launches are chained and overlap, so the rate is bytes per launch over the time the region spends per launch
                "timed_region": None if not alg_bytes else {
                    "ms_per_step": round(step_s * 1e3, 4), "algorithmic_gbs": round(alg_step / step_s / 1e9, 1),
                    "x_hbm_peak": round(alg_step / step_s / 1e9 / HBM_PEAK_GBS, 4),
                    "ray_batches_in_rotation": len(d_sets), "algorithmic_bytes_per_step_mean": int(alg_step),
                    "kernel_event_ms_avg": round(avg_kernel_ms, 4) if avg_kernel_ms else None,
                    "kernel_event_note": "only with --engine-opts '{\"time_kernels\":1}': HIP events around every traversal kernel of the timed region.  Launches are chained "
                                         "lazily: the chain's first kernels (one per lane in rotation) work through all K batches, the later batches are only "
                                         "published to them — so no per-launch duration exists in the timed region; `roofline.kernel_ms_avg` is the isolated one",
                    "fabric_bytes_per_step_chained": pw.get("fabric_bytes_per_step_chained")},
                "limiter": None if not pw else {k: pw.get(k) for k in (
                    "bound", "wave_time_split", "td_busy_frac", "ta_busy_frac", "valu_busy_frac", "valu_lane_util", "salu_share",
                    "l2_hit_rate", "vmem_rd_insts_per_ray", "valu_insts_per_ray", "kernel_ms_isolated", "write_x_compulsory")},
                "limiter_steady_state": steady_state_profile() if (on_profiled_workload and args.workload == "diffuse" and args.quality == 1) else None,
                "profile_source": (prof or {}).get("source"),
                "profile_stale": bool(prof["stale"]) if prof else None})
            # the other configs beside configs[2] (the headline): same definitions
            if on_profiled_workload and args.workload == "diffuse" and "coherent_1M" in extras:
                pc = prof.get("coherent", {})
                keys = ("achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel_ms_avg", "fabric_frac_of_hbm_peak", "l1_gather_frac")
                by = {"configs[2] 1M first-bounce diffuse": {k: roofline.get(k) for k in keys},
                      "configs[1] 1M coherent primaries": roofline_core((golden.get("coherent_1M") or {}).get("bytes"), extras["coherent_1M"]["ms_per_step"], pc.get("fabric_bytes_per_launch"), ceiling)}
                for k, pk in (("configs[2] 1M first-bounce diffuse", pw), ("configs[1] 1M coherent primaries", pc)):
                    if by[k] is not None:
                        by[k]["limiter"] = {q: pk.get(q) for q in ("td_busy_frac", "valu_busy_frac", "valu_lane_util", "l2_hit_rate", "vmem_rd_insts_per_ray", "valu_insts_per_ray")} if pk else None
                if xl:
                    by["battlefield-synth-XL 1M incoherent rays (HBM can bind here)"] = xl.get("xl")
                    by["battlefield-synth-XL 1M first-bounce diffuse (bench camera)"] = xl.get("xl_diffuse")
                extras["roofline_by_config"] = by
            if prof and prof["stale"]:
                print("bench: %s was taken with other kernel sources; re-run tools/profile_bench.sh + tools/summarize_profile.py" % PROFILE_DIR, file=sys.stderr)

        scene_label = ("battlefield-synth-XL" if xl_run else "battlefield-synth") + " (stand-in; reference scene unavailable), %d triangles, " % len(sc["indices"])
        if args.workload == "xl":
            workload_text = scene_label + "1M incoherent rays (uniform origins and directions) per GPU per step — a profiling run, not a BASELINE config"
        elif args.workload == "coherent":
            workload_text = scene_label + "1M coherent primary rays per GPU per step (BASELINE configs[1]), steps issued back to back over %d engine lanes" % ctx.auto_lanes
        elif args.mode == "weak":
            workload_text = scene_label + "1M 1st-bounce diffuse rays per GPU per step (BASELINE configs[2]/[3]), steps issued back to back over %d engine lanes" % ctx.auto_lanes
        else:
            workload_text = scene_label + "ONE 8M-ray 1st-bounce diffuse batch per step cut into %d contiguous shards (BASELINE configs[3])" % world
        line = {
            "metric": "Mrays/s", "value": round(value, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "pre_timed_launches": iso_n * (2 if iso_same_ms else 1),      # isolated launches (the roofline's kernel duration) issued before the warm-up steps: they also bring the clocks up
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.mode, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text,
                       "rays_per_gpu": n, "scene": sc["name"], "parallelism": "rays sharded x%d, scene replicated" % world,
                       "tree": ("racc_host_build_options.quality = %d: %d inner nodes, %d pairs (reference format; " % (args.quality, len(host.nodes), host.pair_count)) +
                               ("the reference's builder, Bvh2.cpp restated)" if args.quality == 0 else "one pair per leaf + re-inserted subtrees; `reference_builder_tree` = the quality-0 tree in the same loop)"),
                       "ray_batches_in_rotation": len(d_sets),
                       "grid_blocks": launch["grid_blocks"], "waves_per_simd": launch["waves_per_simd"], "lanes": lanes,
                       "lanes_in_rotation": ctx.auto_lanes, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            # what the N > 1 figure is made of: every rank's own rate over ITS K steps, and how many ranks the communicator had
            "per_rank_mrays_per_s": [round(n * args.steps / t / 1e6, 1) for t in per_rank] if args.mode == "weak" else [round((total_rays / world) * args.steps / t / 1e6, 1) for t in per_rank],
            "collective_backend": (backend if world > 1 else None), "rccl_ranks": (comm_ranks if (world > 1 and backend == "nccl") else None), "comm_ranks": comm_ranks,
        }
        line.update(extras)
        print(json.dumps(line), flush=True)

    scene.destroy()
    env.destroy()
    ctx.destroy()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
