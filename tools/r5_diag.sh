#!/bin/bash
# stress: racc::render with null callbacks, many short runs under watchdogs; which configuration hangs, and where
cd "${GRAFT_REPO_ROOT:-.}"
python -u - <<'PY'
import sys
sys.path.insert(0, ".")
from rayaccel_amd import synth
synth.write_scene_bin("/tmp/scene1080.bin", synth.battlefield_synth(), viewport=(1920, 1080))
PY
hangs=0
for round in 1 2 3 4 5 6; do
for cfg in "X=1" "RACC_SLICE_ALWAYS=1" "RACC_BATCH=262144" "RACC_GPU_THREADS=6" "RACC_BATCH=262144 RACC_SLICE_ALWAYS=1" "RACC_GPU_THREADS=6 RACC_SLICE_ALWAYS=1"; do
  env RACC_CPU_THREADS=16 RACC_RENDER_WATCHDOG_S=4 RACC_HOST_WATCHDOG_S=6 $cfg timeout -k 5 60 tests/cpp/render_check /tmp/scene1080.bin --null-callbacks 1920 1080 16 6 > /tmp/out.txt 2> /tmp/err.txt
  rc=$?
  if [ $rc -ne 0 ]; then hangs=$((hangs+1)); echo "== round $round [$cfg] rc=$rc"; tail -c 300 /tmp/out.txt; grep -v "amdgpu.ids" /tmp/err.txt | tail -c 1500; else echo "ok round $round [$cfg] $(python -c "import json;d=json.loads(open('/tmp/out.txt').read().strip().splitlines()[-1]);print(d['mrays_per_s_best'], d['mrays_per_s_mean'])")"; fi
done
done
echo "failures: $hangs"
