#!/bin/bash
# One rocprofv3 --pmc pass per counter group over tools/pmc_workload.py (counters in their own runs, --kernel-trace only),
# then the per-workload summary.   usage (on the GPU box):  bash tools/pmc_collect.sh <outdir> [cases]
set -u
OUT=${1:-gpurun_out/pmc}
CASES=${2:-cfg2,cfg3,generic,astero,tree,quantiles,primitives,sampler}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
CGROUPS=(
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum"
)
i=0
for g in "${CGROUPS[@]}"; do
  d="$ROOT/$OUT/pass$i"
  timeout 900 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$d" -- \
      python "$ROOT/tools/pmc_workload.py" --manifest "$ROOT/$OUT/manifest.json" --cases "$CASES" > "$ROOT/$OUT/pass$i.log" 2>&1
  echo "pass $i ($g): rc=$?"; tail -2 "$ROOT/$OUT/pass$i.log"
  i=$((i+1))
done
cd "$ROOT"
python tools/pmc_summarize.py "$OUT/manifest.json" "$OUT" "$OUT/pmc_summary.json"
# keep the raw CSVs out of the merge-back (large): only logs + summary + manifest travel
find "$OUT" -name "*.csv" -size +2M -delete
