#!/bin/bash
# Round profile bundle: rocprofv3 kernel-trace stats of the default bench command + PMC passes of one launch
# + kernel stats of the classification head at the CIFAR-10 shape.  Outputs under gpurun_out/prof_*.
set -u
export TMPDIR=/tmp
R=$PWD
cd /tmp || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1 || { echo bench-stats failed; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cifar -- python3 $R/tools/bench_wide.py --T 20 --reps 1 > $R/gpurun_out/prof_cifar.log 2>&1 || { echo cifar-stats failed; exit 1; }
cd $R && bash tools/profile_pmc.sh gpurun_out/pmc
