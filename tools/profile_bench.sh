#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box).  --kernel-trace/--stats and each --pmc set are SEPARATE runs (never
# combined with sys/hip/hsa traces); every pass has its own timeout (a counter set the hardware cannot collect makes
# rocprofv3 abort and then hang).  Usage: tools/profile_bench.sh <tag>      then: python tools/summarize_profile.py <tag>
# Passes:
#   stats                 the bench command as the driver runs it (chained launches over the lanes): kernel trace + stats
#   stats_one_lane[_X]    the same with one lane and no chaining (every kernel alone, one batch each); X = coherent: configs[1]'s batch; v10: kernel_variant 50
#   pmc_<i>               one counter set per pass, diffuse batch, one lane, no chaining: counters describe ONE launch of ONE batch alone on the GPU
#   pmcc_<i>              the same for the coherent primary batch (--workload coherent), fewer sets
#   stats_one_lane_v10, pmcv_<i>   the compressed 4-wide kernel (kernel_variant 50 = racc::setFastTraversal), one lane, no chaining (round 6)
#   pmcx_<i>              the limiter counter sets in the timed region's OWN mode — default options: three lanes, lazily chained, the instantiation with
#                         in-kernel miss shading (round 6: every set, not only FETCH_SIZE / WRITE_SIZE); rocprofv3 serialises dispatches under --pmc, so
#                         the chain's first kernel does the work of all 20 steps: summarize_profile.py sums over the chain's kernels and divides by 20
#   stats_one_lane_q0, pmcq_<i>, stats_one_lane_xl_q0, pmcxq_<i>   the reference builder's tree (--quality 0; every other pass runs bench.py's default, quality 1)
#   pmcxl_<i>, pmcxd_<i>  battlefield-synth-XL (1.3 GB on the device: past the Infinity Cache), 1M incoherent rays / the camera's 1M diffuse rays, one lane, no chaining
# RACC_BENCH_ISO_LAUNCHES=0: no isolated launches before the warm-up, so the LAST 20 traversal dispatches of every pass are the 20 timed
# steps (tools/summarize_profile.py selects them from the end; round 3's passes picked rows 4..23, which fell inside the isolated block).
export RACC_BENCH_ISO_LAUNCHES=0
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3"
ONE='{"lanes":1,"chain_launches":2}'
V10='{"lanes":1,"chain_launches":2,"kernel_variant":50}'
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_one_lane" -- $CMD --engine-opts "$ONE" > "$OUT/stats_one_lane.log" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_one_lane_coherent" -- $CMD --workload coherent --engine-opts "$ONE" > "$OUT/stats_one_lane_coherent.log" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_one_lane_v10" -- $CMD --engine-opts "$V10" > "$OUT/stats_one_lane_v10.log" 2>&1
pass() {   # pass <dir prefix> <index> "<counters>" <extra bench args...>
  local pre=$1 i=$2 set=$3; shift 3
  timeout -k 5 240 rocprofv3 --pmc $set --output-format csv -d "$OUT/${pre}_$i" -- $CMD "$@" > "$OUT/${pre}_$i.log" 2>&1 || echo "pass ${pre}_$i ($set) failed"
}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1)); pass pmc $i "$set" --engine-opts "$ONE"
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT" "TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
  i=$((i+1)); pass pmcc $i "$set" --workload coherent --engine-opts "$ONE"
done
LIMITER_SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT" "TA_TA_BUSY_sum TD_TD_BUSY_sum")
i=0
for set in "${LIMITER_SETS[@]}"; do
  i=$((i+1)); pass pmcx $i "$set"
done
i=0
for set in "${LIMITER_SETS[@]}"; do
  i=$((i+1)); pass pmcv $i "$set" --engine-opts "$V10"
done
# the reference builder's tree (--quality 0) under the same kernel: what the tree post-processing changes in the counters
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_one_lane_q0" -- $CMD --quality 0 --engine-opts "$ONE" > "$OUT/stats_one_lane_q0.log" 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT" "TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
  i=$((i+1)); pass pmcq $i "$set" --quality 0 --engine-opts "$ONE"
done
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_one_lane_xl" -- $CMD --workload xl --engine-opts "$ONE" > "$OUT/stats_one_lane_xl.log" 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_one_lane_xl_diffuse" -- $CMD --workload xl_diffuse --engine-opts "$ONE" > "$OUT/stats_one_lane_xl_diffuse.log" 2>&1
passxl() {   # the XL scene takes ~20 s to build: its own, longer timeout
  local pre=$1 i=$2 set=$3; shift 3
  timeout -k 5 400 rocprofv3 --pmc $set --output-format csv -d "$OUT/${pre}_$i" -- $CMD "$@" > "$OUT/${pre}_$i.log" 2>&1 || echo "pass ${pre}_$i ($set) failed"
}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
  i=$((i+1)); passxl pmcxl $i "$set" --workload xl --engine-opts "$ONE"
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); passxl pmcxd $i "$set" --workload xl_diffuse --engine-opts "$ONE"
done
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_one_lane_xl_q0" -- $CMD --workload xl --quality 0 --engine-opts "$ONE" > "$OUT/stats_one_lane_xl_q0.log" 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); passxl pmcxq $i "$set" --workload xl --quality 0 --engine-opts "$ONE"
done
find "$OUT" -name "*.csv" | wc -l
du -sh "$OUT"
