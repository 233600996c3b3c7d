#!/bin/bash
# A/B of several BUILDS of libracc_hip.so on one box: tools/ab_builds.sh base v1 v2 ...  swaps rayaccel_amd/libracc_hip_<name>.so in turn (two rounds)
# under tools/gpu_cache_ab.py (single launches, chained sequences) and tools/gpu_chain_k.py (K = 1 ... 64 chained batches); restores the first one.
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do for which in "$@"; do
cp rayaccel_amd/libracc_hip_$which.so rayaccel_amd/libracc_hip.so
echo "== $which"
python tools/gpu_cache_ab.py '{}' 2>&1 | tail -1 | cut -c1-330
python tools/gpu_chain_k.py '{}' 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: v[0] for k, v in d['ms'].items()}, d['fit_best'])"
done; done
cp rayaccel_amd/libracc_hip_$1.so rayaccel_amd/libracc_hip.so
