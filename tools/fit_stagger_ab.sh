#!/bin/bash
# A/B of the staggered schedule (option stagger = output groups on their own streams) against the defaults, plain runs:
# bash tools/fit_stagger_ab.sh > gpurun_out/fit_stagger_ab.txt
set -o pipefail
for cfg in "4096 4" "2048 4" "1024 4" "8192 4" "4096 8" "4096 2" "3072 4"; do
  set -- $cfg
  m=$2
  for opt in "" "stagger=2" "stagger=4" "stagger=2,aggregate=1" "stagger=$m,aggregate=1" "stagger=2,lookahead=0" "stagger=$m,lookahead=0"; do
    BOCF_OPTIONS=$opt timeout -k 5 120 python tools/fit_only.py $1 $m || exit 1
  done
done
