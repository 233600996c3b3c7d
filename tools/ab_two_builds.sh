#!/bin/bash
# A/B of two BUILDS of libracc_hip.so on one box (box-to-box variation is +-2 %, this resolves +-0.3 %): rayaccel_amd/libracc_hip_base.so (e.g. built from
# `git archive HEAD` in /tmp) against libracc_hip_new.so (the working tree), swapped in turn under tools/gpu_cache_ab.py (single launches, chained
# sequences) and tools/gpu_chain_k.py (K = 1 ... 64 chained batches).  Both files are git-ignored and travel with gpurun.
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do for which in base new; do
cp rayaccel_amd/libracc_hip_$which.so rayaccel_amd/libracc_hip.so
echo "== $which"
python tools/gpu_cache_ab.py '{}' 2>&1 | tail -1 | cut -c1-330
python tools/gpu_chain_k.py '{}' 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: v[0] for k, v in d['ms'].items()}, d['fit_best'])"
done; done
cp rayaccel_amd/libracc_hip_new.so rayaccel_amd/libracc_hip.so
