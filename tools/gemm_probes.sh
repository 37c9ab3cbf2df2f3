# Timing-only probes of the variance GEMM (wrong results, option "kstar_valu_probe"): bash tools/gemm_probes.sh
#   two-buffer kernel (swizzle 256): 0 baseline, 2 no loop barriers, 3 no LDS fragment reads, 4 no operand fetch in the loop
set -e
mkdir -p gpurun_out/ab
for v in ${1:-0 2 3 4}; do
  sw=256
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --option swizzle=$sw --option kstar_valu_probe=$v > gpurun_out/ab/v$v.json 2>gpurun_out/ab/v$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/v$v.json").read().strip().splitlines()[-1])
print("probe", $v, "ms_per_step", round(d["ms_per_step"], 3), "gemm_ms", round(d["roofline"].get("kernel_ms", 0), 3), "frac_of_78.6", round(d["roofline"]["frac"], 4), flush=True)
PY
done
