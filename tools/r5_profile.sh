#!/bin/bash
# round 5: the rocprofv3 passes + step statistics, condensed into profiles/r05 on the box (only gpurun_out/ travels back)
cd "${GRAFT_REPO_ROOT:-.}"
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 300 python tools/gpu_step_stats.py gpurun_out/step_stats.json > gpurun_out/step_stats.log 2>&1
bash tools/profile_bench.sh r05 > gpurun_out/profile_r05.log 2>&1
tail -5 gpurun_out/profile_r05.log
python tools/summarize_profile.py r05 > gpurun_out/summarize_r05.log 2>&1; tail -3 gpurun_out/summarize_r05.log
mkdir -p gpurun_out/profiles_r05 && cp -r profiles/r05/* gpurun_out/profiles_r05/
du -sh gpurun_out/prof_r05 gpurun_out/profiles_r05
# the raw pass directories are large: keep only what summarize_profile.py reads
find gpurun_out/prof_r05 -name "*agent_info*" -delete
