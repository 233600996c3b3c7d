#!/bin/bash
# tools/steady_pmc.sh                                   what binds the kernel in STEADY STATE -> gpurun_out/steady_state_pmc.json (below)
# tools/steady_pmc.sh probe <tag> "<counters of pass 1>" "<pass 2>" ...     ad-hoc PMC passes over bench.py's timed launches (RACC_BENCH_ARGS
#   adds bench flags): each pass its own rocprofv3 run (--pmc only, never combined with traces); prints the mean per traversal launch.
if [ "$1" = probe ]; then
shift
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 ${RACC_BENCH_ARGS}"
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 5 180 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- $CMD > "$OUT/p$i.log" 2>&1 || tail -3 "$OUT/p$i.log"
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = {}
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    byc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "traverseKernel" in r["Kernel_Name"]:
            byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in byc.items():
        v.sort(); vals = [x[1] for x in v]
        out[k] = round(sum(vals[4:24]) / max(1, len(vals[4:24])), 1)
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
exit 0
fi
# What binds the kernel in STEADY STATE (round 5).  profiles/<round>/derived.json describes an isolated 1M-ray launch, a third of which is ramp-up and
# drain; here the same counters for 8M-ray launches (bench.py --mode strong at N = 1: one 8M-ray batch per step, one lane, no chaining — a launch
# is 1.65 ms, of which ramp-up and drain are ~8 %).  One rocprofv3 --pmc run per counter set (the probe mode above), condensed with
# tools/summarize_profile.py's formulas into gpurun_out/steady_state_pmc.json (tools/summarize_profile.py copies it into profiles/<round>/).
cd "${GRAFT_REPO_ROOT:-.}"
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export RACC_BENCH_ISO_LAUNCHES=0
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
      "GRBM_GUI_ACTIVE GRBM_COUNT" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
RACC_BENCH_ARGS='--mode strong --engine-opts {"lanes":1,"chain_launches":2}' bash tools/steady_pmc.sh probe steady "${SETS[@]}" > gpurun_out/steady_pmc_raw.json 2>&1
python - <<'PY'
import json
c = json.load(open("gpurun_out/pmc_steady/summary.json"))
rays, CUS, SIMDS, XCDS = 8 << 20, 256, 1024, 8
cycles = c["GRBM_GUI_ACTIVE"] / XCDS
wc = c["SQ_WAVE_CYCLES"]
insts = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
import sys
sys.path.insert(0, ".")
import bench
d = dict(kernel_source_sha256=bench.kernel_source_sha256(), what="traverseKernelV8 (default), 8M first-bounce diffuse rays per launch on the quality tree, the kernel alone on the GPU (one lane, no chaining): "
              "means over the 20 timed launches, one rocprofv3 --pmc run per counter set; formulas as in derived.json",
         kernel_ms=round(cycles / 2.4e6, 4), mrays_per_s=round(rays / (cycles / 2.4e9) / 1e6, 1),
         wave_time_split=dict(waiting=round(c["SQ_WAIT_ANY"] / wc, 4), issue_stalled=round(c["SQ_WAIT_INST_ANY"] / wc, 4), executing=round(c["SQ_ACTIVE_INST_ANY"] / wc, 4)),
         valu_busy_frac=round(4 * c["SQ_ACTIVE_INST_VALU"] / (SIMDS * cycles), 4), valu_lane_util=round(c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]), 4),
         td_busy_frac=round(c["TD_TD_BUSY_sum"] / (CUS * cycles), 4), ta_busy_frac=round(c["TA_TA_BUSY_sum"] / (CUS * cycles), 4),
         salu_share=round(c["SQ_INSTS_SALU"] / insts, 4), valu_insts_per_ray=round(c["SQ_INSTS_VALU"] / rays, 2), vmem_rd_insts_per_ray=round(c["SQ_INSTS_VMEM_RD"] / rays, 3),
         lds_insts_per_ray=round(c["SQ_INSTS_LDS"] / rays, 3), l2_hit_rate=round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4),
         counters=c)
json.dump(d, open("gpurun_out/steady_state_pmc.json", "w"), indent=1)
print(json.dumps({k: v for k, v in d.items() if k != "counters"}, indent=1))
PY
rm -rf gpurun_out/pmc_steady/p*/
