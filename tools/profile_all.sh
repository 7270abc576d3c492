#!/bin/bash
# Round profile bundle: rocprofv3 kernel-trace stats of the default bench command + PMC passes of one launch
# + kernel stats of the classification head at the CIFAR-10 shape and of the hidden-256 streamed kernel + their PMC passes.  Outputs under gpurun_out/prof_*.
set -u
export TMPDIR=/tmp
R=$PWD
cd /tmp || exit 1
# micro-benchmarks behind DESIGN.md section 4 (built by make -C tools/ubench)
{ timeout -k 10 60 $R/tools/ubench/coexec && timeout -k 10 120 $R/tools/ubench/interleave && timeout -k 10 120 $R/tools/ubench/lonewave && timeout -k 10 60 $R/tools/ubench/mfma_order && timeout -k 10 60 $R/tools/ubench/permlane; } > $R/gpurun_out/ubench.txt 2>&1 || { echo ubench failed; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1 || { echo bench-stats failed; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cifar -- python3 $R/tools/bench_wide.py --T 20 --reps 1 > $R/gpurun_out/prof_cifar.log 2>&1 || { echo cifar-stats failed; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stream -- python3 $R/bench.py --workload sweep256 --rows 16384 --nz 128 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_stream.log 2>&1 || { echo stream-stats failed; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mnist -- python3 $R/bench.py --workload mnist --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_mnist.log 2>&1 || { echo mnist-stats failed; exit 1; }
cd $R && bash tools/profile_pmc.sh gpurun_out/pmc &&
bash tools/profile_pmc.sh gpurun_out/pmc_stream --workload sweep256 --rows 2048 --nz 128 --steps 1 --warmup 0 --no-cpu-baseline &&
bash tools/profile_pmc_wide.sh gpurun_out/pmc_wide
