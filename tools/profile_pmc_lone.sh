#!/bin/bash
# PMC of the persistent kernel with ONE wave per SIMD at most (8192 chains, cooperative kernel switched off):
# where does a lone wave's time go?  Usage (GPU box, repo root): bash tools/profile_pmc_lone.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_lone}
R=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
export CARD_AMD_COOP=0
cd /tmp || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d "$R/$OUT/a" -- python3 "$R/bench.py" --rows 64 --nz 128 --steps 1 --warmup 0 --no-cpu-baseline > "$R/$OUT/a.log" 2>&1 || { echo fail a; tail -3 "$R/$OUT/a.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$R/$OUT/b" -- python3 "$R/bench.py" --rows 64 --nz 128 --steps 1 --warmup 0 --no-cpu-baseline > "$R/$OUT/b.log" 2>&1 || { echo fail b; tail -3 "$R/$OUT/b.log"; exit 1; }
echo ok
