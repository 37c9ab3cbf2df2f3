#!/bin/bash
# What does SQ_LDS_BANK_CONFLICT count for DENSE 16-byte LDS accesses?  (the variance GEMM shows conflict cycles =
# 36 % of SQ_LDS_IDX_ACTIVE although its ds_read_b128 / ds_write_b128 are contiguous per 16-lane group.)  Runs the LDS variants of
# tools/mfma_peak_probe.hip -- every lane reads / writes 16 contiguous bytes at lane * 16: no two lanes of a wave ever touch the same bank
# in the same pass -- under the two counters, on the GPU box:   bash tools/lds_conflict_probe.sh > gpurun_out/lds_conflict_probe.txt
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DLDS_ONLY $R/tools/mfma_peak_probe.hip -o /tmp/mfma_peak_lds || exit 1
rm -rf /tmp/lds_pmc
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/lds_pmc -- /tmp/mfma_peak_lds > /tmp/lds_pmc.log 2>&1 || { tail -5 /tmp/lds_pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("/tmp/lds_pmc/*/*counter_collection.csv"))[-1]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    agg.setdefault(k, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
print("%-70s %14s %14s %12s %s" % ("kernel (mfma_lds_loop<NL, KIND, THREADS>: KIND 0/4 dense ds_read_b128, 1 ds_read_b64 @16 B, 2 ds_write_b128, 3 ds_write_b64 @16 B, 5..8 GEMM fragment pattern, rows +2176/+1152/+2112/+2304 B)", "LDS_IDX_ACTIVE", "BANK_CONFLICT", "INSTS_LDS", "conflict / active"))
for k, v in agg.items():
    if "lds_loop" not in k:
        continue
    a, c = v.get("SQ_LDS_IDX_ACTIVE", 0.0), v.get("SQ_LDS_BANK_CONFLICT", 0.0)
    print("%-70s %14.4g %14.4g %12.4g %.3f" % (k[-70:], a, c, v.get("SQ_INSTS_LDS", 0.0), c / a if a else 0.0))
PY
