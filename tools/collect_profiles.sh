#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the rocprofv3 evidence for bench.py's
# default command into gpurun_out/prof_<tag>/ ; tools/summarize_profiles.py then writes profiles/<round>/.
set -o pipefail
TAG=${1:-r02}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
# fit-side evidence: kernel trace + MFMA-busy counters of the Cholesky / inverse GEMMs and the K(X,X) build
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fit_trace -- python3 tools/fit_only.py > $OUT/fit_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/fit_pmc -- python3 tools/fit_only.py > $OUT/fit_pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/fit_pmc_write -- python3 tools/fit_only.py > $OUT/fit_pmc_write.log 2>&1
python3 bench.py --f32 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_f32.json 2> /dev/null
python3 bench.py --config 5 --f32 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_f32.json 2> /dev/null
python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_f64.json 2> /dev/null
BOCF_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
BOCF_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --comm torch > $OUT/bench_rccl_world1_torch.json 2> /dev/null
BOCF_FORCE_DIST=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --shard-fit > $OUT/bench_rccl_world1_shardfit.json 2> /dev/null
python3 bench.py --config 2 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg2.json 2> /dev/null
# the int8 (exact digit products) variance contraction, option predict_i8
python3 bench.py --i8 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_i8.json 2> /dev/null
python3 bench.py --i8 --C 8192 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_i8_c8192.json 2> /dev/null
python3 bench.py --i8 --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_i8.json 2> /dev/null
python3 bench.py --i8 --config 2 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg2_i8.json 2> /dev/null
python3 tools/i8_check.py > $OUT/int8_vs_fp64_by_size.txt 2>&1
python3 tools/i8_small_timing.py > $OUT/int8_small_model_timing.txt 2>&1
python3 bench.py --C 8192 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c8192.json 2> /dev/null
# round 4: fit schedules by size, hyper-parameter update times, config-2 step trace, the HBM-bound kernels variant by variant, team timeline
for n in 256 512 1024 1536 2048 3072 4096 6144; do python3 tools/fit_only.py $n 4; done > $OUT/fit_by_size.txt 2>&1
for n in 256 512 1024 1536 2048 3072 4096; do BOCF_OPTIONS=team_fit=0 python3 tools/fit_only.py $n 4; done > $OUT/fit_by_size_launched.txt 2>&1
for n in 64 128 256 512 1024; do python3 tools/hyper_update.py $n 4 4; done > $OUT/hyper_update_timing.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg2_trace -- python3 bench.py --config 2 --steps 200 --warmup 5 --no-cpu-baseline > $OUT/cfg2_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hyper_trace -- python3 tools/hyper_update.py 256 4 4 > $OUT/hyper_trace.log 2>&1
# stand-alone probes (built in the container, the binaries travel with the snapshot):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hbm_kernel_probe.hip -o tools/hbm_kernel_probe.bin
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ozaki_tile_probe.hip -o tools/ozaki_tile_probe.bin
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_i8_peak.hip -o tools/mfma_i8_peak.bin
[ -x tools/hbm_kernel_probe.bin ] && ./tools/hbm_kernel_probe.bin > $OUT/hbm_kernel_probe.txt 2>&1
[ -x tools/ozaki_tile_probe.bin ] && ./tools/ozaki_tile_probe.bin > $OUT/ozaki_tile_probe.txt 2>&1
[ -x tools/mfma_i8_peak.bin ] && ./tools/mfma_i8_peak.bin > $OUT/mfma_i8_peak.txt 2>&1
BOCF_PROBES=1 BOCF_TEAM_TL=$OUT/team_timeline_N1024.txt python3 tools/team_check.py 1024 > $OUT/team_check.txt 2>&1
python3 tools/team_timeline.py $OUT/team_timeline_N1024.txt > $OUT/team_timeline_N1024_summary.txt 2>&1
for n in 128 1024 4096; do python3 tools/latency_prof.py $n 16; done > $OUT/small_path_latency.txt 2>&1
python3 tools/bo_iteration.py > $OUT/bo_iteration_timing.txt 2>&1
tail -c 600 $OUT/bench.json
