#!/bin/bash
# A/B of two builds of the library over the bench workloads inside one box: tools/ab_workloads.sh [other.so]
# ("base" lines: CARD_AMD_LIB=other.so, "tree" lines: the in-tree library)
set -u
B=${1:-$PWD/card_amd/csrc/_ab/base.so}
for wl in "mnist --steps 10 --warmup 2" "boston --steps 20 --warmup 3" "year --steps 3 --warmup 1" "toy --steps 1 --warmup 1"; do
  for rep in 1 2; do
    CARD_AMD_LIB=$B timeout -k 10 200 python3 bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', '$wl'.split()[0], d['ms_per_step'], round(d['roofline']['frac'],4))" || exit 1
    timeout -k 10 200 python3 bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree', '$wl'.split()[0], d['ms_per_step'], round(d['roofline']['frac'],4))" || exit 1
  done
done
