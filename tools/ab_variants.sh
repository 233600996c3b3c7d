#!/bin/bash
# A/B of engine options on bench.py's own loop: tools/ab_variants.sh '{"kernel_variant":44}' '{}' ...   (K = 20 and 200 each, twice)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for rep in 1 2; do for o in "$@"; do for k in 20 200; do
  python bench.py --steps $k --warmup 5 --no-extras --no-cpu-baseline --engine-opts "$o" 2>/dev/null > /tmp/ab.json
  python - "$o" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print("opts", sys.argv[1], "K", d["steps"], "value", d["value"], "ms_per_step", d["ms_per_step"], flush=True)
PY
done; done; done
