#!/bin/bash
# PMC passes over the classification back end (tools/bench_wide.py, CIFAR-10 head shape, a few timesteps).
# Usage (GPU box, repo root): bash tools/profile_pmc_wide.sh <outdir> [tcc]
set -u
OUT=${1:-gpurun_out/pmc_wide}
R=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
run_pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$R/$OUT/$name" -- python3 "$R/tools/bench_wide.py" --T 6 --reps 1 ${WIDE_ARGS:-} \
      > "$R/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$R/$OUT/$name.log"; exit 1; }
  echo "pass $name ok"
}
if [ "${2:-all}" = "tcc" ]; then          # fabric traffic only
  run_pass tcc_fetch FETCH_SIZE GRBM_GUI_ACTIVE && run_pass tcc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  exit $?
fi
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
run_pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS &&
run_pass tcc_fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run_pass tcc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
