#!/bin/bash
# A/B of the Cholesky schedules: bash tools/fit_sweep.sh > gpurun_out/fit_sweep.log
for N in 1024 2048 4096 8192; do
  for opt in "lookahead=2" "lookahead=1,aggregate=1" "lookahead=0,aggregate=1" "aggregate=2" "aggregate=4"; do
    BOCF_OPTIONS=$opt timeout -k 5 120 python tools/fit_only.py $N 4 || exit 1
  done
done
BOCF_OPTIONS="lookahead=2" timeout -k 5 120 python tools/fit_only.py 4096 1
BOCF_OPTIONS="aggregate=2" timeout -k 5 120 python tools/fit_only.py 4096 1
BOCF_OPTIONS="lookahead=2" timeout -k 5 120 python tools/fit_only.py 8192 8
BOCF_OPTIONS="aggregate=4" timeout -k 5 120 python tools/fit_only.py 8192 8
