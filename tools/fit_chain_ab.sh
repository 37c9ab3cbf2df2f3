#!/bin/bash
# A/B of the persistent-chain schedule (lookahead = 5) against the defaults, plain runs: bash tools/fit_chain_ab.sh > gpurun_out/fit_chain_ab.txt
set -o pipefail
for N in 4096 2048 8192; do
  for opt in "" "lookahead=5" "lookahead=5,overlap_inverse=0" "lookahead=0,aggregate=2" "lookahead=0,aggregate=3"; do
    BOCF_OPTIONS=$opt timeout -k 5 120 python tools/fit_only.py $N 4 || exit 1
  done
done
