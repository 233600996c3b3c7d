#!/bin/bash
# round 5: why the LDS node cache buys nothing — the same counters for the default kernel held to 4 waves per SIMD and for kernel_variant 60
# (524 cached records, 4 waves per SIMD), each kernel alone on the GPU (one lane, no chaining), 1M diffuse rays on the quality tree
cd "${GRAFT_REPO_ROOT:-.}"
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export RACC_BENCH_ISO_LAUNCHES=0
SETS=("SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum")
RACC_BENCH_ARGS='--engine-opts {"lanes":1,"chain_launches":2,"waves_per_simd":4}' bash tools/pmc_probe.sh cache_base4 "${SETS[@]}" > gpurun_out/cache_pmc_base4.json 2>&1
RACC_BENCH_ARGS='--engine-opts {"lanes":1,"chain_launches":2,"kernel_variant":60}' bash tools/pmc_probe.sh cache_v60 "${SETS[@]}" > gpurun_out/cache_pmc_v60.json 2>&1
RACC_BENCH_ARGS='--engine-opts {"lanes":1,"chain_launches":2,"kernel_variant":63}' bash tools/pmc_probe.sh cache_v63 "${SETS[@]}" > gpurun_out/cache_pmc_v63.json 2>&1
tail -30 gpurun_out/cache_pmc_base4.json gpurun_out/cache_pmc_v60.json gpurun_out/cache_pmc_v63.json
rm -rf gpurun_out/pmc_cache_*/p*/
