#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python tools/gpu_chain_vs_single.py 2>&1 | grep -v amdgpu.ids
echo "--- RACC_CHAIN_LAZY=0"
RACC_CHAIN_LAZY=0 timeout 600 python tools/gpu_chain_vs_single.py 2>&1 | grep -v amdgpu.ids | grep chained
