#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box).  --kernel-trace/--stats and each --pmc set are
# SEPARATE runs (never combined with sys/hip/hsa traces).  Usage: tools/profile_bench.sh <tag>
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $set | tr ' ' '+')
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$name" -- $CMD > "$OUT/pmc_$name.log" 2>&1
done
find "$OUT" -name "*.csv" | head -50
du -sh "$OUT"
