#!/bin/bash
# racc::render with callbacks that cost nothing (render_check --null-callbacks, 1080p x 16 = 31M rays per frame): the scheduler's own
# throughput per configuration — rays in flight, submission threads, stream batch size, CPU threads.   tools/gpu_sched_sweep.sh
cd "${GRAFT_REPO_ROOT:-.}"; export GRAFT_REPO_ROOT=$(pwd)
python -u - <<'PY'
import sys
sys.path.insert(0, ".")
from rayaccel_amd import synth
synth.write_scene_bin("/tmp/scene1080.bin", synth.battlefield_synth(), viewport=(1920, 1080))
PY
run() {
  echo -n "$* : "
  env "$@" timeout 120 tests/cpp/render_check /tmp/scene1080.bin --null-callbacks 1920 1080 16 6 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('best %.0f mean %.0f Mrays/s, %d streams of %d' % (d['mrays_per_s_best'], d['mrays_per_s_mean'], d['streams'], d['streamSize']))"
}
run RACC_CPU_THREADS=16
run RACC_CPU_THREADS=16 RACC_IN_FLIGHT=4194304
run RACC_CPU_THREADS=16 RACC_IN_FLIGHT=16777216
run RACC_CPU_THREADS=16 RACC_IN_FLIGHT=33554432
run RACC_CPU_THREADS=16 RACC_GPU_THREADS=2
run RACC_CPU_THREADS=16 RACC_GPU_THREADS=3
run RACC_CPU_THREADS=16 RACC_GPU_THREADS=6
run RACC_CPU_THREADS=16 RACC_GPU_THREADS=8
run RACC_CPU_THREADS=16 RACC_BATCH=262144
run RACC_CPU_THREADS=16 RACC_BATCH=524288 RACC_IN_FLIGHT=16777216
run RACC_CPU_THREADS=16 RACC_BATCH=65536
run RACC_CPU_THREADS=8
run RACC_CPU_THREADS=12
run RACC_CPU_THREADS=24
run RACC_CPU_THREADS=16 RACC_IN_FLIGHT=16777216 RACC_GPU_THREADS=6
run RACC_CPU_THREADS=16 RACC_IN_FLIGHT=16777216 RACC_GPU_THREADS=3
run RACC_CPU_THREADS=16
