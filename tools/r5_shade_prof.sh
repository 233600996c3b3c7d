cd "${GRAFT_REPO_ROOT:-.}"
python - <<'PY'
import sys
sys.path.insert(0, ".")
from rayaccel_amd import synth
synth.write_scene_bin("/tmp/s1080.bin", synth.battlefield_synth(), viewport=(1920, 1080))
PY
for rep in 1 2 3; do for sc in 0 1; do
echo -n "RACC_PT_SCALAR=$sc: "
RACC_PT_SCALAR=$sc RACC_PROFILE=1 RACC_BUILD_QUALITY=1 python - <<'PY' 2>&1 | grep "RayAccelerator profile: render" | cut -c24-200
import os, sys
sys.path.insert(0, ".")
from rayaccel_amd.engine import path_trace
path_trace("/tmp/s1080.bin", 1920, 1080, 0, 8, device=0, shading="cpu", cpu_threads=16)
PY
done; done
