#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_quality.py -m gpu -x -q > gpurun_out/r5i_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5i_pytest.log
timeout 600 python tools/gpu_sizes.py '{}' '{"drain_prefetch":2}' '{"drain_prefetch":1}' '{}' '{"drain_prefetch":2}' 2>&1 | grep -v amdgpu.ids
for opts in '{}' '{"drain_prefetch":2}' '{}' '{"drain_prefetch":2}'; do
  for k in 20; do
    timeout 300 python bench.py --steps $k --warmup 5 --no-extras --no-cpu-baseline --engine-opts "$opts" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$opts', $k, d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
  done
done
timeout 300 python tools/gpu_small_streams.py 2>&1 | grep -v amdgpu.ids | head -3
