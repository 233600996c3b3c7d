#!/bin/bash
# round 5, third GPU session: the scheduler-only figure (every run under its own timeout), then the bench line
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -u - <<'PY'
import sys, os, subprocess
sys.path.insert(0, ".")
from rayaccel_amd import synth
sc = synth.battlefield_synth()
synth.write_scene_bin("/tmp/scene1080.bin", sc, viewport=(1920, 1080))
print("scene written", flush=True)
PY
for cfg in "" "RACC_SLICE_ALWAYS=1" "RACC_BUILD_QUALITY=1" "RACC_GPU_THREADS=2" "RACC_GPU_THREADS=6" "RACC_BATCH=262144" "RACC_BATCH=524288 RACC_IN_FLIGHT=8388608" "RACC_CPU_THREADS=4" "RACC_PROFILE=1"; do
  echo "== null-callbacks [$cfg]"
  env RACC_CPU_THREADS=16 $cfg timeout -k 5 90 tests/cpp/render_check /tmp/scene1080.bin --null-callbacks 1920 1080 16 4 2>&1 | tail -4
  echo "rc=$?"
done
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05c_k20.json 2> gpurun_out/bench_r05c_k20.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05c_k20.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("reference_builder_tree", {}).get("mrays_per_s_same_loop_as_value"), d.get("batch_scaling"), json.dumps(d.get("path_tracer_1080p"))[:1200])
PY
