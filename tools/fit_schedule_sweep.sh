#!/bin/bash
# Which factorization schedule wins where (plain runs): single stream (lookahead=0), reserved-CU chain (lookahead=2).
# bash tools/fit_schedule_sweep.sh > gpurun_out/fit_schedule_sweep.txt
set -o pipefail
for N in 1024 1536 2048 2560 3072 3584 4096; do
  for m in 1 2 4 8; do
    for opt in "lookahead=0" "lookahead=2"; do
      BOCF_OPTIONS=$opt timeout -k 5 120 python tools/fit_only.py $N $m | python -c "
import sys,re
l=sys.stdin.read()
m=re.search(r'N=(\d+) m=(\d+) \[(.*?)\].*cholesky ([\d.]+), inverse ([\d.]+).*schedule (\d)', l)
print('N=%s m=%s %-24s chol+inv %.3f ms (schedule %s)' % (m.group(1), m.group(2), m.group(3), float(m.group(4))+float(m.group(5)), m.group(6)))
" || exit 1
    done
  done
done
