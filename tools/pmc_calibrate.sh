#!/bin/bash
# Calibrate the memory-side PMC counters on kernels with exactly known byte counts (tools/pmc_calibrate.hip).
#   usage (GPU box):  bash tools/pmc_calibrate.sh <outdir>
# One rocprofv3 --pmc pass per counter group (counters in their own runs, --kernel-trace only); the summary lists,
# per probe kernel, counter value per launch, expected bytes, and the factor bytes / count.
set -u
OUT=${1:-gpurun_out/pmc_cal}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
BIN="$ROOT/tools/pmc_calibrate.bin"
[ -x "$BIN" ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-value "$ROOT/tools/pmc_calibrate.hip" -o "$BIN"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$ROOT/$OUT/counters_available.txt" 2>&1
grep -i -E "^\s*(Name|counter)?.*(TCC_EA|MALL|HBM|DRAM|FETCH|WRITE_SIZE|TCC_REQ|TCC_READ|TCC_WRITE)" "$ROOT/$OUT/counters_available.txt" | head -200 > "$ROOT/$OUT/counters_memory_side.txt"
CGROUPS=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE"
)
i=0
for g in "${CGROUPS[@]}"; do
  d="$ROOT/$OUT/pass$i"
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$d" -- "$BIN" 6 > "$ROOT/$OUT/pass$i.log" 2>&1
  echo "pass $i ($g): rc=$?"
  i=$((i+1))
done
cd "$ROOT"
python tools/pmc_calibrate_summary.py "$OUT" > "$OUT/calibration.txt" 2>&1
cat "$OUT/calibration.txt"
find "$OUT" -name "*.csv" -size +2M -delete
