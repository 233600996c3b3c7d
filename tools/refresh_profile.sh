#!/bin/bash
# From the build container: re-take profiles/<tag> on an MI355X box (gpurun), condense it, and keep the bench line of the same tree.
TAG=${1:-r03}
cd "$(dirname "$0")/.."
rm -rf gpurun_out/prof_$TAG
gpurun --timeout 2400 -- "tools/profile_bench.sh $TAG 2>&1 | tail -3; python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -c 300 gpurun_out/bench_$TAG.json" 2>&1 | tail -5
python tools/summarize_profile.py $TAG | grep -v formulas
python -m pytest tests/test_profiles.py -q 2>&1 | tail -1
