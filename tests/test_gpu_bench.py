"""GPU tests of bench.py itself: the one-line JSON contract at N=1 and the N>1 flow (rehearsed with gloo on one GPU: the
ranks share device 0; the real run is nccl = RCCL, one rank per GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline"}


def _line(cmd, env=None):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    # gloo's own chatter (rehearsal only; two ranks interleave it, so a fragment may lack the "[Gloo]" prefix)
    lines = [l for l in p.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]") and "connected peer ranks" not in l]
    assert len(lines) == 1, "bench must print exactly one line on stdout: %r" % lines[-3:]
    return json.loads(lines[0])


def test_single_gpu_line():
    d = _line([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--grid", "128", "--no-extras", "--gather"])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["unit"] == "Mrays/s"
    assert d["value"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)        # the measurement contract's keys
    # round 6: `bound` names what binds on this cache-resident scene — the CU gather path, against its MEASURED ceiling — and no `frac` exceeds 1;
    # the HBM yard-stick is carried beside it (hbm_algorithmic_frac may exceed 1: the bytes never cross the fabric, hbm_traffic_frac says how few do)
    assert r["bound"] == "cu_gather_path" and r["unit"] == "GB/s" and r["kernel_ms_avg"] > 0 and 15000 < r["peak"] < 40000
    assert r["algorithmic_bytes_per_launch"] > 0 and abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1 and r["timed_region"]["ms_per_step"] == d["ms_per_step"]
    assert r["hbm_peak_gbs"] == 8000.0 and abs(r["hbm_algorithmic_frac"] - r["achieved"] / 8000.0) < 1e-3

    def fracs(o, path="line"):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "frac" and v is not None:
                    assert v <= 1.0, "%s.frac = %r" % (path, v)
                fracs(v, path + "." + k)
    fracs(d)
    assert d["cpu_baseline"]["kind"] in ("port", "simd-port") and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    # the scene is built exactly as racc::createScene builds it: racc_host_scene_build without options = the library default (quality 1)
    assert "racc::createScene" in d["config"]["scene_build"] and d["config"]["tree"].startswith("quality 1:")
    import rayaccel_amd as ra
    from rayaccel_amd import synth
    sc = synth.battlefield_synth(grid=128, boxes=128 * 6, quads=128 * 28)
    default = ra.HostScene(sc["vertices"], sc["indices"], quality=None)
    assert default.quality == 1 and ("%d inner nodes, %d pairs" % (len(default.nodes), default.pair_count)) in d["config"]["tree"]
    # the K steps with the all-gather of every step's hit records, serialised and overlapped — at N = 1 RCCL with one rank (the plumbing)
    g = d["with_allgather_of_results_overlapped"]
    assert "error" not in g, g
    assert g["ranks"] == 1 and g["mrays_per_s"] > 0 and g["own_shard_in_the_gathered_array_equals_the_timed_results"] is True
    assert d["with_allgather_of_results"]["mrays_per_s"] > 0 and g["bytes_gathered_per_step"] == 16 << 20


def test_other_scene_classes_and_a_scene_file(tmp_path):
    """bench.py --scene city-synth / soup-synth and --scene-file (the reference's .bin layout, Renderer/main.cpp:117-191: where the real battlefield.bin
    drops in): the line is produced, every timed record is held to the oracle inside the run (bench.py refuses to print otherwise)."""
    from rayaccel_amd import synth
    for extra in (["--scene", "city-synth"], ["--scene", "soup-synth"]):
        d = _line([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--grid", "96", "--no-extras"] + extra)
        assert d["value"] > 0 and extra[1] in d["config"]["scene"] and d["config"]["node_visits_per_ray"] > 1 and d["cpu_baseline"]["value"] > 0
    path = os.path.join(str(tmp_path), "scene.bin")
    synth.write_scene_bin(path, synth.battlefield_synth(grid=64, boxes=100, quads=400))
    d = _line([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-extras", "--scene-file", path])
    assert d["value"] > 0 and d["config"]["scene"] == "file:scene.bin" and d["data"].startswith("scene file")


def test_two_ranks_rehearsal():
    env = dict(os.environ, RACC_BENCH_BACKEND="gloo", RACC_BENCH_DEVICE="0")
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29543", "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--grid", "128", "--no-extras"], env)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["cpu_baseline"] is None      # the CPU legs run at N=1 only
    assert d["config"]["parallelism"].startswith("rays sharded x2")


def test_strong_scaling_mode_two_ranks_rehearsal():
    """BASELINE configs[3] as written: ONE 8M-ray batch per step cut into N contiguous shards (bench.py --mode strong),
    rehearsed with two gloo ranks on GPU 0 (the RCCL gather step needs one GPU per rank and is skipped here; it runs on
    hardware with a world of one rank in test_config3_8M_rays_in_8_shards)."""
    env = dict(os.environ, RACC_BENCH_BACKEND="gloo", RACC_BENCH_DEVICE="0")
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29547", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--grid", "128", "--no-extras", "--mode", "strong"], env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["rays_per_gpu"] == 4 << 20 and "contiguous shards" in d["config"]["workload"]


def test_plain_command_with_two_gpus_spawns_its_own_ranks():
    """Round-3 verdict: `python bench.py --gpus N` started as a plain command exited with an error; if the driver's scaling run
    mirrors its N=1 command, every N > 1 point would have failed before it started.  Now the plain form re-executes itself under
    torch.distributed.run (rehearsed with gloo, both ranks on GPU 0); the line says how many ranks the collective had and every
    rank's own rate."""
    env = dict(os.environ, RACC_BENCH_BACKEND="gloo", RACC_BENCH_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    d = _line([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--grid", "128", "--no-extras"], env)
    assert d["n_gpus"] == 2 and d["comm_ranks"] == 2 and d["collective_backend"] == "gloo" and d["rccl_ranks"] is None
    assert len(d["per_rank_mrays_per_s"]) == 2 and all(v > 0 for v in d["per_rank_mrays_per_s"])
    assert d["value"] <= sum(d["per_rank_mrays_per_s"]) * 1.001          # whole-job rate = all rays / the slowest rank's time


def test_a_rank_that_never_joins_the_final_gather_does_not_silence_rank_0():
    """SCALE-day hardening (round-4 verdict, item 7): if the all-gather of the ranks' elapsed times hangs (a rank died, the fabric stalls),
    rank 0 prints its line with "partial": true after RACC_BENCH_GATHER_TIMEOUT seconds instead of hanging silently.  Two gloo ranks on
    GPU 0; rank 1 is told never to enter the gather."""
    env = dict(os.environ, RACC_BENCH_BACKEND="gloo", RACC_BENCH_DEVICE="0", RACC_BENCH_TEST_HANG="1", RACC_BENCH_GATHER_TIMEOUT="8")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29551", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--grid", "128", "--no-extras"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:] + p.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["partial"] is True and d["n_gpus"] == 2 and d["value"] > 0 and "did not return" in d["partial_reason"]
