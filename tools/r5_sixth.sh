#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for lazy in 1 0 1 0; do
  for k in 20 200; do
    RACC_CHAIN_LAZY=$lazy timeout 300 python bench.py --steps $k --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lazy=$lazy', $k, d['value'], d['ms_per_step'])"
  done
done
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r5f_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r5f_pytest.log
