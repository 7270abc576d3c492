#!/bin/bash
# A/B: LDS-resident layer 1 vs hoisted layer 1, one full-size launch per step.
set -u
for xd in 1 13 90; do
  for h in 0 1; do
    [ $xd = 90 ] && [ $h = 0 ] && continue
    v=$(CARD_AMD_HOIST=$h timeout -k 10 200 python3 bench.py --rows 1024 --nz 1024 --x-dim $xd --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4e sample-steps/s  %.1f TFLOP/s  frac %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac']))") || { echo "x_dim $xd hoist $h FAILED"; exit 1; }
    echo "x_dim=$xd hoist=$h  $v"
  done
done
