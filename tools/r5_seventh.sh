#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for rep in 1 2; do
  for k in 20 200; do
    timeout 300 python bench.py --steps $k --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window4', $k, d['value'], d['ms_per_step'])"
  done
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reuse.py tests/test_gpu_quality.py -m gpu -x -q -k "chain or reuse or recycled or quality or interleave" > gpurun_out/r5g_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5g_pytest.log
GRAFT_REPO_ROOT=$(pwd) timeout 400 bash tools/trace_bench.sh > gpurun_out/trace_r05_lazy.txt 2>&1; grep -c " T start" gpurun_out/trace_r05_lazy.txt; tail -30 gpurun_out/trace_r05_lazy.txt | cut -c1-90
