#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python tools/gpu_sizes.py '{}' '{"refill_min":12,"leaf_min":10,"inner_reps":3}' '{}' '{"refill_min":12,"leaf_min":10,"inner_reps":3}' 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/gpu_small_streams.py 2>&1 | grep -v amdgpu.ids | tail -12
