#!/bin/bash
# A/B of the staggered schedule (option stagger = output groups on their own streams) against the defaults, plain runs:
# bash tools/fit_stagger_ab.sh > gpurun_out/fit_stagger_ab.txt
set -o pipefail
for cfg in "2560 2" "2560 4" "2560 8" "3072 2" "3072 3" "3072 4" "3072 8" "3584 2" "3584 4" "3584 8" "4096 2" "4096 3" "4096 4" "4096 6" "5120 2" "5120 4" "6144 2" "6144 4" "2048 2" "2048 8"; do
  set -- $cfg
  for opt in "" "stagger=2,lookahead=0"; do
    BOCF_OPTIONS=$opt timeout -k 5 120 python tools/fit_only.py $1 $2 | cut -c1-200 || exit 1
  done
done
