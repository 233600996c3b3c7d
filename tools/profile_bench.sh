#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box).  --kernel-trace/--stats and each --pmc set are SEPARATE runs (never
# combined with sys/hip/hsa traces); every pass has its own timeout (a counter set the hardware cannot collect makes
# rocprofv3 abort and then hang).  Usage: tools/profile_bench.sh <tag>      then: python tools/summarize_profile.py <tag>
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3"
# (1) the bench command as the driver runs it (chained launches over the lanes) and (2) the same with one lane and no chaining (every kernel alone, one batch each)
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_one_lane" -- $CMD --engine-opts '{"lanes":1,"chain_launches":2}' > "$OUT/stats_one_lane.log" 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  # counters describe ONE launch of ONE batch alone on the GPU with its full grid: one engine lane, launches not chained (rocprofv3 serialises dispatches for --pmc anyway)
  timeout -k 5 240 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$i" -- $CMD --engine-opts '{"lanes":1,"chain_launches":2}' > "$OUT/pmc_$i.log" 2>&1 || echo "pass $i ($set) failed"
done
find "$OUT" -name "*.csv" | wc -l
du -sh "$OUT"
