#!/bin/bash
# A/B kernel builds: one full-size launch each (bench.py --rows 1024 --nz 1024), 2 rounds interleaved.
set -u
ARGS="--rows 1024 --nz 1024 --steps 3 --warmup 1 --no-cpu-baseline ${EXTRA:-}"
for round in 1 2; do
  for so in card_amd/variants/*.so; do
    v=$(CARD_AMD_LIB=$PWD/$so timeout -k 10 120 python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4e  frac %.4f' % (d['value'], d['roofline']['frac']))") || { echo "$so FAILED"; exit 1; }
    echo "round $round  $(basename $so)  $v"
  done
done
