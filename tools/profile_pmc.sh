#!/bin/bash
# PMC passes for the dominant kernel (one counter group per rocprofv3 run, no tracing domains mixed in).
# Usage (on the GPU box, from the repo root):  bash tools/profile_pmc.sh <outdir> [bench args...]
# Workload: ONE full-size launch (1024 rows x 1024 z = 1 048 576 chains = 256 CUs x 8 waves x 16 tiles x 32,
# T=1000), the same per-launch shape as bench.py's default C2 run.
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${*:---rows 1024 --nz 1024 --steps 1 --warmup 0 --no-cpu-baseline}
R=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
run_pass() {
  name=$1; shift
  timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d "$R/$OUT/$name" -- python3 "$R/bench.py" $ARGS \
      > "$R/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$R/$OUT/$name.log"; exit 1; }
  echo "pass $name ok"
}
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
run_pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU &&
run_pass tcc_fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run_pass tcc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
