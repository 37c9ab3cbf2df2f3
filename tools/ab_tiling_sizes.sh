# 128-row vs 256-row (three-buffer) variance GEMM at several batch sizes: bash tools/ab_tiling_sizes.sh
set -e
mkdir -p gpurun_out/ab
for C in 4096 8192 16384 32768 65536; do
  for s in 0 258; do
    python bench.py --C $C --steps 5 --warmup 2 --no-cpu-baseline --option swizzle=$s > gpurun_out/ab/c${C}_s$s.json 2>gpurun_out/ab/c${C}_s$s.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab/c${C}_s$s.json").read().strip().splitlines()[-1])
print("C", $C, "swizzle", $s, "ms_per_step", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 4), flush=True)
PY
  done
done
python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --option swizzle=0 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 swizzle 0', d['ms_per_step'], d['roofline']['frac'])"
python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --option swizzle=258 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 swizzle 258', d['ms_per_step'], d['roofline']['frac'])"
