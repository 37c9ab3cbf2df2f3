# A/B of the variance GEMM's tilings on the headline workload: bash tools/ab_tilings.sh "0 256 257 258"  (option "swizzle" values)
set -e
mkdir -p gpurun_out/ab
for s in ${1:-0 256 257 258}; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --option swizzle=$s > gpurun_out/ab/s$s.json 2>gpurun_out/ab/s$s.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/s$s.json").read().strip().splitlines()[-1])
print($s, d["ms_per_step"], d["roofline"]["frac"], flush=True)
PY
done
