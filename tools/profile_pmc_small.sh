#!/bin/bash
# PMC passes (SQ counters only) for the small-batch kernels: usage  bash tools/profile_pmc_small.sh <outdir> <bench args...>
set -u
OUT=$1; shift
ARGS=$*
R=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
run_pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$R/$OUT/$name" -- python3 "$R/bench.py" $ARGS \
      > "$R/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$R/$OUT/$name.log"; exit 1; }
  echo "pass $name ok"
}
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
run_pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
