#!/bin/bash
# A/B of two builds of the library inside one box: tools/ab_lib.sh <base.so> <program> [args...]
# runs the program alternately with CARD_AMD_LIB=<base.so> and with the in-tree library, three times each.
set -u
base=$1; shift
for rep in 1 2 3; do
  echo "base:"; CARD_AMD_LIB=$base timeout -k 10 200 "$@" || exit 1
  echo "tree:"; timeout -k 10 200 "$@" || exit 1
done
