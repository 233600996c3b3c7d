# Round 3's closing GPU run: the whole -m gpu suite, the microbenchmarks, the rocprofv3 passes of profiles/r03, the bench lines.
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_gputest_final.log | head -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 tools/microbench/run_microbench.sh r03 > gpurun_out/r3_microbench_final.log 2>&1; tail -2 gpurun_out/r3_microbench_final.log
timeout 1500 tools/profile_bench.sh r03 2>&1 | tail -3
python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_r03.err
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03_k20.json 2>/dev/null; echo "bench k20 rc=$?"
python -c "
import json
for f in ('gpurun_out/bench_r03.json','gpurun_out/bench_r03_k20.json'):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d.get('compressed_wide_kernel_variant_50',{}).get('mrays_per_s_same_loop_as_value'))"
