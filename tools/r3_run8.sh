cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=20:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"; done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=200:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
