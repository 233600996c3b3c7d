cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest6.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_gputest6.log
timeout 600 tools/microbench/run_microbench.sh r03 > gpurun_out/r3_microbench6.log 2>&1; tail -3 gpurun_out/r3_microbench6.log
timeout 1500 tools/profile_bench.sh r03 2>&1 | tail -4
python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench_r03.json; tail -3 gpurun_out/bench_r03.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r03_k20.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_r03_k20.json')); print('K=20:', d['value'], d['ms_per_step'], d.get('compressed_wide_kernel_variant_50',{}).get('mrays_per_s_same_loop_as_value'))"
