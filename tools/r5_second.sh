#!/bin/bash
# round 5, second GPU session: suite, step statistics from the GPU, policy sweep on the quality tree, fetch calibration, scheduler-only figure
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r5b_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r5b_pytest.log
tail -4 gpurun_out/r5b_pytest.log
timeout 600 python tools/gpu_step_stats.py gpurun_out/step_stats.json > gpurun_out/step_stats.log 2>&1; tail -8 gpurun_out/step_stats.log
timeout 900 python tools/gpu_policy_sweep.py > gpurun_out/policy_sweep_q1.log 2>&1; echo sweep rc=$?
timeout 600 bash tools/microbench/run_microbench.sh r05 > gpurun_out/microbench_r05.log 2>&1; tail -30 gpurun_out/microbench_r05.log
python - <<'PY'
import sys, os, subprocess, json, tempfile
sys.path.insert(0, ".")
from rayaccel_amd import synth
sc = synth.battlefield_synth()
f = tempfile.NamedTemporaryFile(suffix=".bin", delete=False); f.close()
synth.write_scene_bin(f.name, sc, viewport=(1920, 1080))
for env in ({}, {"RACC_SLICE_ALWAYS": "1"}, {"RACC_BUILD_QUALITY": "1"}, {"RACC_GPU_THREADS": "2"}, {"RACC_GPU_THREADS": "8"}, {"RACC_BATCH": "262144"}, {"RACC_BATCH": "1048576", "RACC_IN_FLIGHT": "8388608"}):
    p = subprocess.run(["tests/cpp/render_check", f.name, "--null-callbacks", "1920", "1080", "16", "4"], capture_output=True, text=True, env=dict(os.environ, RACC_CPU_THREADS="16", **env))
    print("null-callbacks", env, p.stdout.strip()[-400:], p.stderr[-200:])
os.unlink(f.name)
PY
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05b_k20.json 2> gpurun_out/bench_r05b_k20.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05b_k20.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("reference_builder_tree", {}).get("mrays_per_s_same_loop_as_value"), d.get("batch_scaling"), json.dumps(d.get("path_tracer_1080p"))[:900])
PY
