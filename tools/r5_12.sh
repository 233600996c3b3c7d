#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for n in 3 1 2 3 1; do
  for k in 20 200; do
    RACC_CHAIN_KERNELS=$n timeout 300 python bench.py --steps $k --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain kernels $n', $k, d['value'], d['ms_per_step'])"
  done
done
