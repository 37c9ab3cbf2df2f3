#!/bin/bash
# A/B of the Cholesky schedules: bash tools/fit_sweep.sh > gpurun_out/fit_sweep.log
for N in 1024 2048 3072 4096 6144 8192; do
  for opt in "aggregate=1,lookahead=0" "aggregate=2" "aggregate=3" "aggregate=4" "lookahead=2"; do
    BOCF_OPTIONS=$opt timeout -k 5 120 python tools/fit_only.py $N 4 || exit 1
  done
done
