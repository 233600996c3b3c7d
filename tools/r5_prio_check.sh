#!/bin/bash
# round 5, wave priorities (s_setprio in racc_kernel_v8_hot.inc): the GPU suite on the build with them, then battlefield-synth-XL's chained bench
# loop with rayaccel_amd/libracc_hip_base.so (the build without) and libracc_hip_prio.so swapped in turn — where HBM binds the priorities buy
# nothing (1,817-1,819 vs 1,762-1,814 Mrays/s), which is what a bandwidth-bound kernel should show.
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/s6_prio_suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" gpurun_out/s6_prio_suite.log | tail -1
for rep in 1 2; do for which in base prio; do
cp rayaccel_amd/libracc_hip_$which.so rayaccel_amd/libracc_hip.so
echo "== $which"
timeout 300 python bench.py --workload xl --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xl chained', d['value'], d['ms_per_step'], 'iso', (d.get('roofline') or {}).get('kernel_ms_avg'))"
done; done
cp rayaccel_amd/libracc_hip_prio.so rayaccel_amd/libracc_hip.so
