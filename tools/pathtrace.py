#!/usr/bin/env python
"""BASELINE.json configs[4]: path tracer, W x H x spp, multi-bounce ray streams, end to end on N GPUs.

One process per GPU; samples are sharded across ranks (rank r renders samples [r*spp/N, (r+1)*spp/N) of every pixel —
no exchange during rendering) and the partial frame buffers are summed with ONE RCCL reduce at the end (the only step of
this workload that really exchanges data).  Prints one JSON line on rank 0 and optionally writes a PFM image.

    python tools/pathtrace.py --width 1920 --height 1080 --spp 64 [--out frame.pfm]
    python tools/pathtrace.py --gpus 8 --spp 64            # re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nproc-per-node 8 ... tools/pathtrace.py --spp 64
"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=0, help="0 = from the scene file (5 for battlefield-synth)")
    ap.add_argument("--grid", type=int, default=700)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--scene", default=None, help="reference-format .bin (default: battlefield-synth stand-in)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--shading", default="gpu", choices=("gpu", "cpu"), help="gpu: device-resident consumer (pt_device.hip); "
                    "cpu: spawn/shade callbacks on host threads as in the reference (pathtracer.cpp).  Same image either way.")
    ap.add_argument("--batch", type=int, default=0, help="samples per wavefront batch (gpu shading); 0 = 8")
    ap.add_argument("--gpus", type=int, default=0, help="started as a plain command with --gpus N > 1: re-executes itself under torch.distributed.run, one rank per GPU")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    import numpy as np
    import torch
    import torch.distributed as dist
    import rayaccel_amd as ra
    from rayaccel_amd import synth
    from rayaccel_amd.shard import shard_range
    from rayaccel_amd.engine import path_trace
    # RACC_BENCH_BACKEND=gloo + RACC_BENCH_DEVICE=0 rehearse the N>1 flow on a 1-GPU box (ranks share GPU 0, frames are summed
    # on the CPU); the real thing is nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("RACC_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("RACC_BENCH_DEVICE", local))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    scene_file = args.scene
    tmp = None
    if scene_file is None:
        sc = synth.battlefield_synth() if args.grid == 700 else synth.battlefield_synth(grid=args.grid, boxes=args.grid * 6, quads=args.grid * 28)
        tmp = tempfile.NamedTemporaryFile(suffix=".bin", delete=False)
        tmp.close()
        synth.write_scene_bin(tmp.name, sc, viewport=(args.width, args.height))
        scene_file, name = tmp.name, sc["name"] + " (stand-in; reference scene unavailable)"
    else:
        name = scene_file
    first, last = shard_range(args.spp, rank, world)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if last > first:
        img, st = path_trace(scene_file, args.width, args.height, first, last - first, device=local, max_depth=args.depth, cpu_threads=args.threads,
                             shading=args.shading, samples_per_batch=args.batch)
    else:       # more ranks than samples: this rank has none and contributes a black frame (rendering "at least one" would
        import numpy as np      # duplicate another rank's sample and make the image depend on N)
        img, st = np.zeros((args.height, args.width, 3), np.float64), dict(rays_traced=0, seconds=0.0, primary_rays=0, threads=0, max_depth=0, tiles_x=0, tiles_y=0, triangles=0, reserved=0)
    dev = "cuda" if backend == "nccl" or world == 1 else "cpu"
    rays = torch.tensor([float(st["rays_traced"]), st["seconds"]], dtype=torch.float64, device=dev)
    frame = torch.from_numpy(img).to(dev)
    if world > 1:
        dist.reduce(frame, 0, op=dist.ReduceOp.SUM)                 # RCCL: the one exchange step (W*H*3 doubles per rank)
        tot = rays.clone(); dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        mx = rays.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        total_rays, render_s = float(tot[0]), float(mx[1])
    else:
        total_rays, render_s = float(rays[0]), float(rays[1])
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if tmp:
        os.unlink(tmp.name)
    if rank == 0:
        mean = (frame / args.spp).cpu().numpy()
        import hashlib
        digest = hashlib.md5(np.ascontiguousarray(frame.cpu().numpy()).tobytes()).hexdigest()      # of the SUM of radiance: equal for any N
        if args.out:
            with open(args.out, "wb") as f:
                f.write(b"PF\n%d %d\n-1.0\n" % (args.width, args.height))
                f.write(mean[::-1].astype("<f4").tobytes())
        print(json.dumps({"metric": "Mrays/s (path tracer end to end, %s)" % ("generation + shading kernels on the GPU" if args.shading == "gpu" else "host shading included"),
                          "shading": args.shading, "value": round(total_rays / render_s / 1e6, 1),
                          "unit": "Mrays/s", "n_gpus": world, "rays_traced": int(total_rays), "render_seconds": round(render_s, 3),
                          "wall_seconds_incl_scene_build": round(wall, 3), "spp": args.spp, "width": args.width, "height": args.height,
                          "tiles": [st["tiles_x"], st["tiles_y"]], "max_depth": st["max_depth"], "shade_threads_per_rank": st["threads"],
                          "mean_luminance": float(mean.mean()), "frame_md5": digest, "data": "synthetic", "scene": name}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
