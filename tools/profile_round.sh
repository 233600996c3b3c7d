#!/bin/bash
# One round's measurement run on the GPU box, condensed into profiles/<tag> THERE (only gpurun_out/ travels back: the result is copied to
# gpurun_out/profiles_<tag>/): microbenchmarks (the gather ceiling bench.py's roofline is held against), wave-level step statistics, the rocprofv3
# passes over bench.py (tools/profile_bench.sh), the steady-state counters (tools/steady_pmc.sh), tools/summarize_profile.py.
#   usage: tools/profile_round.sh r06
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
bash tools/microbench/run_microbench.sh $TAG > gpurun_out/microbench_$TAG.log 2>&1; tail -2 gpurun_out/microbench_$TAG.log
timeout 300 python tools/gpu_step_stats.py gpurun_out/step_stats.json > gpurun_out/step_stats.log 2>&1
bash tools/profile_bench.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -5 gpurun_out/profile_$TAG.log
bash tools/steady_pmc.sh > gpurun_out/steady_pmc.log 2>&1; tail -3 gpurun_out/steady_pmc.log
python tools/summarize_profile.py $TAG > gpurun_out/summarize_$TAG.log 2>&1; tail -3 gpurun_out/summarize_$TAG.log
mkdir -p gpurun_out/profiles_$TAG && cp -r profiles/$TAG/* gpurun_out/profiles_$TAG/
# the raw pass directories are large: keep only what summarize_profile.py reads
find gpurun_out/prof_$TAG -name "*agent_info*" -delete
du -sh gpurun_out/prof_$TAG gpurun_out/profiles_$TAG
