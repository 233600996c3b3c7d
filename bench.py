#!/usr/bin/env python
"""Headline benchmark: Mrays/s of the intersect-batch hot path on MI355X.

A "step" is one pass of the path over one batch: 1,048,576 first-bounce diffuse
(incoherent) rays on the battlefield-synth stand-in scene per GPU (BASELINE.json
configs[2]; at N GPUs every rank traces its own 1M-ray batch = configs[3]'s 8M rays at
N=8, weak scaling, no data-path collective — rays never interact).  Inputs are resident
in HBM before the timed region.  Output: ONE JSON line on rank 0 (see README/DESIGN.md).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED_GBS = 6290.0      # same guide: float4-copy ceiling
RAYS_PER_BATCH = 1 << 20
KERNEL_NAME = "traverseKernelV8"


def _committed_traffic():
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (or None)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (profiling runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra measurements (profiling runs)")
    ap.add_argument("--grid", type=int, default=700, help="height-field resolution of battlefield-synth (700 = full)")
    ap.add_argument("--engine-opts", default="", help="JSON dict of racc_hip_options overrides (kernel A/B and profiling runs only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    import numpy as np
    import torch                      # first: the engine then shares torch's HIP runtime in this process
    import torch.distributed as dist

    import rayaccel_amd as ra
    from rayaccel_amd import synth

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
    sc = synth.battlefield_synth() if full else synth.battlefield_synth(grid=args.grid, boxes=args.grid * 6, quads=args.grid * 28)
    host = ra.HostScene(sc["vertices"], sc["indices"])
    engine_opts = json.loads(args.engine_opts) if args.engine_opts else {}
    ctx = ra.Context(device=device, **engine_opts)
    scene = ctx.upload_scene(host.nodes, host.pairs, host.remap)
    env = ctx.create_environment(sc["env"])

    primary, _ = synth.primary_rays(sc["camera"], 1024, 1024)
    primary_hits = ctx.intersect(scene, env, primary)                       # GPU path, host buffers
    bounce = synth.diffuse_bounce_rays(sc, primary, primary_hits, RAYS_PER_BATCH, first_sample=rank)
    n = len(bounce)

    d_rays = torch.from_numpy(bounce.view(np.float32).reshape(n, 8).copy()).cuda()
    d_out = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def run(iters, rays_t=d_rays, out_t=d_out):
        return ctx.intersect_device_timed(scene, env, rays_t.data_ptr(), out_t.data_ptr(), rays_t.shape[0], iters)

    if args.warmup:
        run(args.warmup)

    # ---- timed region: exactly K steps, barrier + device sync on both sides --------------------
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = run(args.steps)          # K back-to-back launches, a HIP event pair around each, on the launch stream
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    value = world * n * args.steps / elapsed / 1e6
    launch = ctx.launch_info()

    # ---- optional extras, all outside the timed region ------------------------------------------
    extras = {}
    if world > 1 and backend == "nccl":   # RCCL all-gather of the Result shards over xGMI (only needed by a GPU-side consumer)
        gathered = torch.empty((world * n, 4), dtype=torch.float32, device="cuda")
        dist.all_gather_into_tensor(gathered, d_out)
        torch.cuda.synchronize(); barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            dist.all_gather_into_tensor(gathered, d_out)
        torch.cuda.synchronize()
        extras["allgather_results_ms"] = round((time.perf_counter() - t1) / 5 * 1e3, 4)
        extras["allgather_bytes"] = int(gathered.numel() * 4)
    if rank == 0 and not args.no_extras:
        d_prim = torch.from_numpy(primary.view(np.float32).reshape(len(primary), 8).copy()).cuda()
        d_prim_out = torch.zeros((len(primary), 4), dtype=torch.float32, device="cuda")
        run(2, d_prim, d_prim_out)
        pm = float(np.median(run(10, d_prim, d_prim_out)))
        extras["coherent_1M"] = {"ms_per_step": round(pm, 4), "mrays_per_s": round(len(primary) / pm / 1e3, 1)}
        # The reference keeps gpuSubmissionThreads (4) streams in flight (RayAccelerator.cpp:436,711-717).  Same K steps
        # issued round-robin over 4 lanes (HIP streams): launches overlap, so one launch's drain hides under the next
        # one's bulk.  Reported beside `value`, which stays the one-launch-at-a-time figure the roofline is quoted on.
        outs = [torch.zeros((n, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
        for lane in range(4):
            ctx.intersect_device(scene, env, d_rays.data_ptr(), outs[lane].data_ptr(), n, lane=lane)
        for lane in range(4):
            ctx.wait(lane)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            ctx.intersect_device(scene, env, d_rays.data_ptr(), outs[k % 4].data_ptr(), n, lane=k % 4)
        for lane in range(4):
            ctx.wait(lane)
        torch.cuda.synchronize()
        extras["four_streams_in_flight_mrays_per_s"] = round(args.steps * n / (time.perf_counter() - t1) / 1e6, 1)
        if not torch.equal(outs[0].view(torch.int32), d_out.view(torch.int32)):   # bit compare (a miss id reads as NaN in f32)
            sys.exit("bench: overlapped launches changed the results")
        # PCIe-inclusive rate of the host-buffer entry point (never `value`)
        res_host = np.zeros(n, ra.RESULT_DTYPE)
        ctx.intersect(scene, env, bounce, res_host)
        t1 = time.perf_counter()
        for _ in range(3):
            ctx.intersect(scene, env, bounce, res_host)
        extras["host_buffers_pcie_inclusive_mrays_per_s"] = round(3 * n / (time.perf_counter() - t1) / 1e6, 1)
        # the same with the two host arrays page-locked (racc_hip_register_host, what racc::createContext does with its
        # stream block): the copies go by DMA and big batches are sliced so that copies run beside kernels
        lib = ra.load_library()
        ray_host = np.ascontiguousarray(bounce)
        if lib.racc_hip_register_host(ctx._h, ray_host.ctypes.data, ray_host.nbytes) == 0 and \
                lib.racc_hip_register_host(ctx._h, res_host.ctypes.data, res_host.nbytes) == 0:
            ctx.intersect(scene, env, ray_host, res_host)
            t1 = time.perf_counter()
            for _ in range(5):
                ctx.intersect(scene, env, ray_host, res_host)
            extras["host_buffers_page_locked_mrays_per_s"] = round(5 * n / (time.perf_counter() - t1) / 1e6, 1)
            if not np.array_equal(res_host["triangle"], d_out.cpu().numpy().view(ra.RESULT_DTYPE).reshape(-1)["triangle"]):
                sys.exit("bench: the sliced host-buffer path changed the results")
            lib.racc_hip_unregister_host(ctx._h, ray_host.ctypes.data)
            lib.racc_hip_unregister_host(ctx._h, res_host.ctypes.data)

        # Batch-size scaling of the traversal kernel (same diffuse rays, 8 sample sets): T(N) = fixed + per-ray cost.
        if world == 1 and full:
            many = np.concatenate([bounce] + [synth.diffuse_bounce_rays(sc, primary, primary_hits, RAYS_PER_BATCH, first_sample=k) for k in range(1, 8)])
            d_many = torch.from_numpy(many.view(np.float32).reshape(len(many), 8).copy()).cuda()
            d_many_out = torch.zeros((len(many), 4), dtype=torch.float32, device="cuda")
            scaling = {}
            for nn in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 23):
                ctx.intersect_device_timed(scene, env, d_many.data_ptr(), d_many_out.data_ptr(), nn, 2)
                scaling[str(nn)] = round(float(np.median(ctx.intersect_device_timed(scene, env, d_many.data_ptr(), d_many_out.data_ptr(), nn, 7))), 4)
            slope = (scaling[str(1 << 23)] - scaling[str(1 << 20)]) / 7.0          # ms per 2^20 rays
            extras["batch_scaling"] = {"kernel_ms_by_rays": scaling, "steady_state_mrays_per_s": round((1 << 20) / slope / 1e3, 1),
                                       "fixed_ms": round(scaling[str(1 << 20)] - slope, 4)}
            del d_many, d_many_out

        # BASELINE configs[4]: the path tracer, 1920x1080, end to end on this GPU.  Device-resident consumer (generation and
        # shading kernels around racc_hip_intersect_device) at 64 spp; the reference-shaped consumer (spawn/shade callbacks on
        # host threads through racc::render, PCIe both ways) at 8 spp.  Both render the same image (tests/test_gpu_pathtracer.py).
        if world == 1 and full:
            import tempfile
            from rayaccel_amd.engine import path_trace
            tmp = tempfile.NamedTemporaryFile(suffix=".bin", delete=False)
            tmp.close()
            try:
                synth.write_scene_bin(tmp.name, sc, viewport=(1920, 1080))
                _, sg = path_trace(tmp.name, 1920, 1080, 0, 64, device=device, shading="gpu")
                _, sh = path_trace(tmp.name, 1920, 1080, 0, 8, device=device, shading="cpu", cpu_threads=usable_cores())
                extras["path_tracer_1080p"] = {
                    "gpu_shading_64spp": {"mrays_per_s": round(sg["rays_traced"] / sg["seconds"] / 1e6, 1), "seconds": round(sg["seconds"], 4), "rays": int(sg["rays_traced"])},
                    "host_shading_8spp": {"mrays_per_s": round(sh["rays_traced"] / sh["seconds"] / 1e6, 1), "seconds": round(sh["seconds"], 4), "rays": int(sh["rays_traced"]),
                                          "shade_threads": int(sh["threads"])}}
            finally:
                os.unlink(tmp.name)

    # ---- roofline + CPU baseline (rank 0) --------------------------------------------------------
    roofline, cpu_baseline = None, None
    if rank == 0:
        avg_kernel_ms = float(np.mean(kernel_ms))
        alg_bytes, src = None, None
        if not args.no_cpu_baseline and world == 1:       # the CPU legs run at N=1 only
            from oracle import oracle            # checker / CPU leg only; never on the product path
            blobs = host.blobs()
            ref, nv, npairs, _ = oracle.traverse(blobs, bounce, env=sc["env"], counters=True)
            alg_bytes, src = oracle.algorithmic_bytes(ref, nv, npairs), "oracle counters, live"
            got = d_out.cpu().numpy().view(ra.RESULT_DTYPE).reshape(-1)
            if not np.array_equal(got["triangle"], ref["triangle"]):
                sys.exit("bench: GPU results differ from the oracle — refusing to report a number")
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
                            "sample": "the full 1,048,576-ray diffuse batch, %d passes per timing x 3 timings (median), %d pthreads x "
                                      "1024-ray slices; CPU BVH2 restatement standing in for Embree (Embree unavailable)" % (repeat, threads)}
            # The reference's OWN traversal kernel (oracle/_ref, built from RayAccelerator/Kernels.h with its own flags) on this
            # same GPU and batch, launched as the reference launches it (work-groups of 8, enqueue + clFinish).
            try:
                from oracle import ref_kernel
                if ref_kernel.built():
                    ref_res, ref_t = ref_kernel.run(blobs, bounce, sc["env"], repeats=5)
                    hit = ref["triangle"] != 0xFFFFFFFF
                    agree = float((ref_res["triangle"][hit] == ref["triangle"][hit]).mean())
                    extras["reference_opencl_kernel_on_this_gpu"] = {
                        "mrays_per_s": round(n / float(np.median(ref_t)) / 1e6, 1), "ms_per_launch": round(float(np.median(ref_t)) * 1e3, 3),
                        "primId_agreement_with_engine": round(agree, 6),
                        "what": "Kernels.h `traversal`, -cl-fast-relaxed-math, local size 8, same 1M-ray diffuse batch, enqueue + clFinish"}
            except Exception as e:   # noqa: BLE001 - a missing OpenCL runtime must not fail the bench
                extras["reference_opencl_kernel_on_this_gpu"] = {"error": str(e)[:200]}
        else:
            try:
                with open(os.path.join(ROOT, "tests", "golden", "algorithmic_bytes.json")) as f:
                    alg_bytes = json.load(f)["diffuse_1M_sample0"]["bytes"] if full else None
                    src = "tests/golden/algorithmic_bytes.json"
            except (OSError, KeyError, ValueError):
                pass
        if alg_bytes:
            achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
            traffic = _committed_traffic()
            roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic.get("hbm_bytes_per_launch") if traffic else None,
                        "kernel": KERNEL_NAME, "kernel_ms_avg": round(avg_kernel_ms, 4),
                        "algorithmic_bytes_per_launch": int(alg_bytes), "algorithmic_bytes_source": src,
                        "frac_of_measured_6290": round(achieved / HBM_MEASURED_GBS, 4)}

        line = {
            "metric": "Mrays/s", "value": round(value, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "battlefield-synth (stand-in; reference scene unavailable), %d triangles, "
                                   "1M 1st-bounce diffuse rays per GPU (BASELINE configs[2]/[3])" % len(sc["indices"]),
                       "rays_per_gpu": n, "scene": sc["name"], "parallelism": "rays sharded x%d, scene replicated" % world,
                       "grid_blocks": launch["grid_blocks"], "waves_per_simd": launch["waves_per_simd"]},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
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
