#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for opts in '{}' '{"kernel_variant":44}' '{"tail_active":48}' '{"kernel_variant":44,"tail_active":48}' '{}' '{"kernel_variant":44}'; do
  for k in 20 200; do
    timeout 300 python bench.py --steps $k --warmup 5 --no-extras --no-cpu-baseline --engine-opts "$opts" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$opts', $k, d['value'], d['ms_per_step'])"
  done
done
