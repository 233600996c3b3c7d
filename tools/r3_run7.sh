cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; echo "bench rc=$?"; tail -c 400 gpurun_out/bench_r03.json; tail -3 gpurun_out/bench_r03.err
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03_k20.json 2>/dev/null; echo "bench k20 rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/bench_r03_k20.json')); print('K=20:', d['value'], d['ms_per_step'], d.get('compressed_wide_kernel_variant_50',{}).get('mrays_per_s_same_loop_as_value')); print(json.dumps(d['roofline'])[:1500])"
