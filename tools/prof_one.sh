#!/bin/bash
# rocprofv3 kernel-trace stats of one command: tools/prof_one.sh <name> <program> [args...]  -> gpurun_out/prof_<name>/ + a printed summary
set -u
export TMPDIR=/tmp
R=$PWD
name=$1; shift
cd /tmp || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -- "$@" > $R/gpurun_out/prof_$name.log 2>&1 || { echo "prof $name failed"; tail -5 $R/gpurun_out/prof_$name.log; exit 1; }
cd $R
f=$(find gpurun_out/prof_$name -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls %6s avg %12.1f us  %6.2f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
