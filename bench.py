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
    ... bench.py --mode strong      # BASELINE configs[3] as written: ONE 8M-ray batch cut into N contiguous shards
    ... bench.py --scene city-synth | soup-synth | --scene-file some.bin      # another scene class / a reference-format scene file (Renderer/main.cpp:117-191)
The line also carries (tools/bench_extras.py): `roofline` (the measured roof: the CU gather path on this cache-resident scene, HBM on
battlefield-synth-XL), `cpu_baseline`, and — at any N — the same K steps with the RCCL all-gather of every step's hit records, serialised
and overlapped (`with_allgather_of_results[_overlapped]`).
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

RAYS_PER_BATCH = 1 << 20
NSETS = 8                      # ray batches that rotate through the steps of a timed region (sample sets of the same primary hits)
XL_RAY_SEED = 7
KERNEL_NAME = "traverseKernelV8"
# rocprofv3 summaries of THIS command (tools/profile_bench.sh <round>) + microbenchmark outputs: the latest round's that is committed
PROFILE_DIR = next(os.path.join("profiles", r) for r in ("r06", "r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", r, "derived.json")))
# what the committed counters depend on: the kernel, its launch policy (chunk, grid, chain), the device node order — and the tree builder
KERNEL_SOURCES = ("rayaccel_amd/csrc/racc_kernel_v8.inc", "rayaccel_amd/csrc/racc_kernel_v8_hot.inc", "rayaccel_amd/csrc/racc_device.inc", "rayaccel_amd/csrc/racc_launch.inc",
                  "rayaccel_amd/csrc/racc_scene_format.inc", "rayaccel_amd/csrc/racc_hip.hip", "rayaccel_amd/csrc/scene_build.cpp")


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
    """The limiter counters of 8M-ray launches (tools/steady_pmc.sh -> profiles/<round>/steady_state_pmc.json): what the kernel does with its
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
    ap.add_argument("--no-gather", action="store_true", help="skip the all-gather figures (`with_allgather_of_results[_overlapped]`); RACC_BENCH_NO_GATHER=1 does the same")
    ap.add_argument("--gather", action="store_true", help="the all-gather figures even with --no-extras (they are part of the default line since round 6)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle check and the CPU legs (profiling runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra measurements (profiling runs)")
    ap.add_argument("--workload", choices=("diffuse", "coherent", "xl", "xl_diffuse"), default="diffuse",
                    help="diffuse (default, the headline): 1M first-bounce diffuse rays per step (configs[2]); coherent: the 1M primary rays (configs[1]); "
                         "xl / xl_diffuse: battlefield-synth-XL (25 M triangles, 1.5 GB on the device: past the Infinity Cache) with 1M incoherent rays / "
                         "its camera's 1M first-bounce diffuse rays — profiling runs of those configs")
    ap.add_argument("--scene", choices=("battlefield-synth", "city-synth", "soup-synth"), default="battlefield-synth",
                    help="scene class (rayaccel_amd/synth.py); battlefield-synth is the stand-in every BASELINE config is measured on")
    ap.add_argument("--scene-file", default=None, help="a scene file in the reference's format (Renderer/main.cpp:117-191: the real battlefield.bin drops in here); overrides --scene")
    ap.add_argument("--grid", type=int, default=700, help="size knob of the synthetic scene: height-field resolution of battlefield-synth (700 = full; the other classes scale with it)")
    ap.add_argument("--quality", type=int, default=None, choices=(0, 1, 2),
                    help="racc_host_build_options.quality of the scene build.  Default: NOT passed — the scene is built exactly as racc::createScene builds it "
                         "(racc_host_scene_build with no options: the library default, quality 1 since round 6).  0 = the reference's builder (Bvh2.cpp restated, "
                         "byte-identical to the oracle's); the line reports that tree beside the headline either way (`reference_builder_tree`)")
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

    import types
    import numpy as np
    import torch                      # first: the engine then shares torch's HIP runtime in this process
    import torch.distributed as dist

    import rayaccel_amd as ra
    from rayaccel_amd import synth
    from rayaccel_amd.shard import shard_range
    from tools import bench_extras as bx

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
    standard_scene = args.scene_file is None and args.scene == "battlefield-synth"
    if args.scene_file:
        sc = synth.read_scene_bin(args.scene_file)
        sc["name"] = "file:" + os.path.basename(args.scene_file)
    elif xl_run:
        sc = synth.battlefield_synth_xl() if full else synth.battlefield_synth_xl(grid=args.grid)
    elif args.scene == "city-synth":
        sc = synth.city_synth() if full else synth.city_synth(blocks=max(4, args.grid // 8))
    elif args.scene == "soup-synth":
        sc = synth.soup_synth() if full else synth.soup_synth(triangles=2 * args.grid * args.grid, clusters=max(4, args.grid // 8))
    else:
        sc = synth.battlefield_synth() if full else synth.battlefield_synth(grid=args.grid, boxes=args.grid * 6, quads=args.grid * 28)
    # (N ranks share the host's cores while they set up: the BVH build of every rank takes its share, not all of them)
    os.environ.setdefault("RACC_BUILD_THREADS", str(max(1, usable_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))))
    # quality=None: racc_host_scene_build without options — byte for byte the build racc::createScene performs (tests/test_gpu_bench.py asserts it)
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
    # works through many batches or finds nothing left, so no per-launch duration exists.  Untimed, before the warm-up; 60 launches, the mean
    # of the last 30 (the duration settles only after the GPU has been busy for ~15 ms), first re-tracing one batch, then rotating through the
    # sample sets as the timed steps and the committed rocprofv3 passes do.  Reported as `pre_timed_launches` (RACC_BENCH_ISO_LAUNCHES=0 skips
    # them — the profiling passes do, so that their last K traversal dispatches are the K steps).
    # No collector pause inside the K steps or between the warm-up and them: a full collection of this process takes ~35 ms, during
    # which the GPU would sit idle and drop its clocks.  Collected here, once; switched back on after the timed region.
    gc.collect()
    gc.disable()
    iso_n = int(os.environ.get("RACC_BENCH_ISO_LAUNCHES", "60"))
    iso_ms = None
    iso_same_ms = None
    if iso_n > 0:
        iso_all = ctx.intersect_device_timed(scene, env, d_rays.data_ptr(), outs[-1].data_ptr(), n, iso_n)
        iso_ms = float(np.mean(iso_all[len(iso_all) // 2:]))
        if len(d_sets) > 1:
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

    def emergency_line(reason):
        """What rank 0 prints when a collective does not come back: the figure from ITS OWN elapsed time, marked partial."""
        return {"metric": "Mrays/s", "value": round(total_rays * args.steps / elapsed / 1e6, 1), "unit": "Mrays/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": args.mode, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "partial": True, "partial_reason": reason,
                "config": {"workload": "battlefield-synth (stand-in), 1M 1st-bounce diffuse rays per GPU per step", "rays_per_gpu": n}}

    per_rank, comm_ranks = [elapsed], 1
    if world > 1:
        # SCALE-day hardening: rank 0's line must not depend on another rank's teardown.  If the gather of the elapsed times has not come back
        # after RACC_BENCH_GATHER_TIMEOUT seconds (default 120; a rank died, the fabric hangs), rank 0 prints a line from ITS OWN elapsed time,
        # marked "partial": true, and the process exits — instead of hanging until the driver's limit with nothing on stdout.
        import threading
        gather_timeout = float(os.environ.get("RACC_BENCH_GATHER_TIMEOUT", "120"))

        def give_up():
            if rank == 0:
                print(json.dumps(emergency_line("the all-gather of the ranks' elapsed times did not return within %.0f s: value = all ranks' rays over RANK 0's own time, "
                                                "not the maximum over ranks" % gather_timeout)), flush=True)
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
    S = types.SimpleNamespace(
        args=args, ctx=ctx, scene=scene, env=env, sc=sc, host=host, n=n, world=world, rank=rank, device=device, lanes=lanes, barrier=barrier,
        total_rays=total_rays, bounce=bounce, ray_sets=ray_sets, d_sets=d_sets, outs=outs, primary=primary, primary_hits=primary_hits,
        full=full, xl_run=xl_run, standard_scene=standard_scene, elapsed=elapsed, iso_ms=iso_ms, iso_same_ms=iso_same_ms, iso_n=iso_n, kernel_ms=kernel_ms,
        usable_cores=usable_cores, committed_profile=committed_profile, steady_state_profile=steady_state_profile, gather_ceiling=gather_ceiling,
        profile_dir=PROFILE_DIR, kernel_name=KERNEL_NAME, golden={},
        d_ref_bits=outs[0].view(torch.int32).clone(),      # the default kernel's records of the timed batch (the extras reuse the result arrays)
        set_bits=[outs[k].view(torch.int32).clone() for k in range(1, min(len(d_sets), args.steps, len(outs)))])      # ... and of the other sample sets

    scene_label = ("battlefield-synth-XL" if xl_run else sc["name"].split("(")[0]) + (" (stand-in; reference scene unavailable)" if not args.scene_file else "") + ", %d triangles, " % len(sc["indices"])
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
        "dtype": "f32", "data": "synthetic" if not args.scene_file else "scene file, synthetic rays",
        "config": {"workload": workload_text,
                   "rays_per_gpu": n, "scene": sc["name"], "parallelism": "rays sharded x%d, scene replicated" % world,
                   "scene_build": ("racc_host_scene_build without options = what racc::createScene performs (library default: quality %d)" % host.quality) if args.quality is None
                                  else "racc_host_scene_build_ex(quality = %d), asked for on the command line" % args.quality,
                   "tree": ("quality %d: %d inner nodes, %d pairs (reference format; " % (host.quality, len(host.nodes), host.pair_count)) +
                           ("the reference's builder, Bvh2.cpp restated)" if host.quality == 0 else "built over triangle references (spatial splits), one pair per leaf, subtrees re-inserted; `reference_builder_tree` = the quality-0 tree in the same loop)"),
                   "ray_batches_in_rotation": len(d_sets),
                   "grid_blocks": launch["grid_blocks"], "waves_per_simd": launch["waves_per_simd"], "lanes": lanes,
                   "lanes_in_rotation": ctx.auto_lanes, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
        "roofline": None, "cpu_baseline": None,
        # what the N > 1 figure is made of: every rank's own rate over ITS K steps, and how many ranks the communicator had
        "per_rank_mrays_per_s": [round(n * args.steps / t / 1e6, 1) for t in per_rank] if args.mode == "weak" else [round((total_rays / world) * args.steps / t / 1e6, 1) for t in per_rank],
        "collective_backend": (backend if world > 1 else None), "rccl_ranks": (comm_ranks if (world > 1 and backend == "nccl") else None), "comm_ranks": comm_ranks,
    }

    # ---- extras, all outside the timed region (tools/bench_extras.py) ----------------------------
    extras = {}
    want_gather = (args.gather or not (args.no_gather or args.no_extras or os.environ.get("RACC_BENCH_NO_GATHER"))) and (backend == "nccl" or world == 1) and not xl_run
    if want_gather:
        # The same K steps with the all-gather of every step's hit records, serialised and overlapped.  A collective on every
        # rank: under a watchdog of its own at N > 1 — if the C-ABI's RCCL instance does not come up on some rank, rank 0 still prints the line
        # (without these figures) and every rank leaves.
        import threading
        wd = None
        # (RCCL prints a version banner on STDOUT when its first communicator comes up: stdout is pointed at stderr for the duration — the
        #  line is the only thing this command prints there)
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        if world > 1:
            def give_up_gather():
                if rank == 0:
                    line["with_allgather_of_results_overlapped"] = {"error": "the all-gather extras did not finish within %s s: line printed without them" % os.environ.get("RACC_BENCH_GATHER_EXTRA_TIMEOUT", "180")}
                    os.write(real_stdout, (json.dumps(line) + "\n").encode())
                os._exit(0)      # (every rank leaves quietly: the line is out, the launcher must not report a failed job)
            wd = threading.Timer(float(os.environ.get("RACC_BENCH_GATHER_EXTRA_TIMEOUT", "180")), give_up_gather)
            wd.daemon = True
            wd.start()
        try:
            extras.update(bx.allgather_extras(S))
        except Exception as e:   # noqa: BLE001 - librccl missing / refusing must not cost the line
            extras["with_allgather_of_results_overlapped"] = {"error": str(e)[:300]}
        if wd:
            wd.cancel()
        sys.stdout.flush()
        import ctypes
        ctypes.CDLL(None).fflush(None)      # (the banner sits in C stdio's buffer — a pipe is fully buffered — and would come out at exit, behind the line)
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    # (N > 1: no other untimed extras — rank 0 would still be measuring while the other ranks tear the process group down)
    if rank == 0 and world == 1 and not args.no_extras:
        extras.update(bx.single_gpu_extras(S))
    xl = None
    if rank == 0 and world == 1 and full and standard_scene and args.mode == "weak" and args.workload == "diffuse" and not args.no_extras:
        xl = bx.xl_rooflines(S)

    # ---- oracle check, roofline + CPU baseline (rank 0) -----------------------------------------
    if rank == 0:
        alg_bytes, alg_by_set, src = None, None, None
        try:
            with open(os.path.join(ROOT, "tests", "golden", "algorithmic_bytes.json")) as f:
                S.golden = json.load(f)
        except (OSError, ValueError):
            pass
        if host.quality:
            S.golden = S.golden.get("quality%d" % host.quality, {})
        if not args.no_cpu_baseline and world == 1 and args.mode == "weak":       # the CPU legs run at N=1 only
            alg_bytes, alg_by_set, src, line["cpu_baseline"], more, visits = bx.oracle_check_and_cpu_legs(S)
            extras.update(more)
            line["config"].update(visits)
        elif full and args.mode == "weak" and standard_scene:
            golden_key = {"diffuse": "diffuse_1M_sample0", "coherent": "coherent_1M", "xl": "xl_1M", "xl_diffuse": "xl_diffuse_1M"}[args.workload]
            alg_bytes, src = (S.golden.get(golden_key) or {}).get("bytes"), "tests/golden/algorithmic_bytes.json"
        line["roofline"] = bx.build_roofline(S, alg_bytes, alg_by_set, src, extras, xl)
        line.update(extras)
        print(json.dumps(line), flush=True)

    scene.destroy()
    env.destroy()
    ctx.destroy()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
