#!/bin/bash
# round 5, fourth GPU session: timeline of the driver's 20 steps, focused policy sweep on the quality tree, XL chained, render stress
cd "${GRAFT_REPO_ROOT:-.}"
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 400 bash tools/trace_bench.sh > gpurun_out/trace_r05.txt 2>&1; tail -45 gpurun_out/trace_r05.txt
timeout 600 python tools/gpu_policy_sweep.py '{}' '{"tail_active":48}' '{"tail_active":56}' '{"tail_active":48,"refill_min":8}' '{"tail_active":48,"refill_min":16}' '{"tail_active":48,"leaf_min":8}' '{"tail_active":48,"leaf_min":14}' '{"tail_active":48,"inner_reps":4}' '{"tail_active":48,"thin_reps":4}' '{"tail_active":48,"thin_reps":16}' '{"tail_active":40,"leaf_min":8,"refill_min":10}' '{}' > gpurun_out/policy_sweep2_q1.log 2>&1; cat gpurun_out/policy_sweep2_q1.log
timeout 600 python tools/gpu_xl_sweep.py '{}' '{"tail_active":48}' > gpurun_out/xl_sweep_q1.log 2>&1; cat gpurun_out/xl_sweep_q1.log
timeout 600 python -m pytest tests/test_gpu_render.py -x -q 2>&1 | tail -3
