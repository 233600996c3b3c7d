#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python tools/gpu_chain_vs_single.py 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
  for k in 20 200; do
    timeout 300 python bench.py --steps $k --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queued env', $k, d['value'], d['ms_per_step'])"
  done
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reuse.py tests/test_gpu_quality.py -m gpu -x -q -k "chain or reuse or recycled or quality or interleave" > gpurun_out/r5h_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r5h_pytest.log
