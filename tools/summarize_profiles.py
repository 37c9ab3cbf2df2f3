"""Summarise gpurun_out/prof_<tag>/ (from tools/collect_profiles.sh) into profiles/<tag>/."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench_cfg3.json"))
def newest(pattern):      # gpurun merges every call's files into the same directory: take the latest run's
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]
ks = newest(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join(dst, "bench_cfg3_kernel_stats.csv"))
pmc = {}
for d in ("pmc_fetch", "pmc_write", "pmc_mfma", "pmc_sq"):
    for f in newest(os.path.join(src, d, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, x in v.items():
                pmc.setdefault(k, {})[c] = {"dispatches": len(x), "mean": sum(x) / len(x), "max": max(x)}
json.dump(pmc, open(os.path.join(dst, "bench_cfg3_pmc.json"), "w"), indent=1, sort_keys=True)
b = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1])
cfg = b["config"]
gk = [k for k in pmc if "gemm_tn_f64_sumsq256" in k or "gemm_tn_f64_kernel<1" in k][0]
g = pmc[gk]
import subprocess
try:
    head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"]).decode().strip()
except Exception:
    head = "unknown"
import hashlib
_src = open(os.path.join(root, "bocf_amd", "csrc", "gemm_f64.hip"), "rb").read()
source_sha = hashlib.sha1(b"blob %d\0" % len(_src) + _src).hexdigest()      # = git hash-object: bench.py prices traffic only against the same source
traffic = {"N": cfg["N"], "m": cfg["m"], "C_local": cfg["C"] // b["n_gpus"], "kernel": gk, "git_head": head, "source_sha": source_sha,
           "FETCH_SIZE_KiB": g["FETCH_SIZE"]["mean"], "WRITE_SIZE_KiB": g["WRITE_SIZE"]["mean"],
           "note": "separate --pmc passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)"}
json.dump(traffic, open(os.path.join(dst, "gemm_traffic.json"), "w"), indent=1)
t = g["GRBM_GUI_ACTIVE"]["mean"] / 8
print("gemm: clock-cycles %.4g  mfma busy frac %.3f  traffic %.1f GB" % (
    t, g["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (1024 * t), (2 * traffic["FETCH_SIZE_KiB"] + traffic["WRITE_SIZE_KiB"]) * 1024 / 1e9))
print(open(ks).read()[:1500])
for extra in ("bench_f32.json", "bench_cfg2.json", "bench_c8192.json", "bench_cfg5_f32.json", "bench_cfg5_f64.json", "bench_rccl_world1.json",
              "bench_rccl_world1_torch.json", "bench_rccl_world1_shardfit.json", "bench_i8.json", "bench_i8_c8192.json", "bench_cfg5_i8.json", "bench_cfg2_i8.json"):
    if os.path.exists(os.path.join(src, extra)):
        lines = [l for l in open(os.path.join(src, extra)) if l.startswith("{")]      # (RCCL prints its banner to stdout in front of the line)
        if lines:
            open(os.path.join(dst, extra), "w").write(lines[-1])
fk = newest(os.path.join(src, "fit_trace", "*", "*kernel_stats.csv"))
if fk:
    shutil.copy(fk[0], os.path.join(dst, "fit_N4096_m4_kernel_stats.csv"))
fit = {}
for d in ("fit_pmc", "fit_pmc_write"):
    for f in newest(os.path.join(src, d, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in agg.items():
            for c, x in v.items():
                fit.setdefault(k, {})[c] = {"dispatches": len(x), "sum": sum(a for a, _ in x), "total_ns": sum(b for _, b in x)}
if fit:
    json.dump(fit, open(os.path.join(dst, "fit_N4096_m4_pmc.json"), "w"), indent=1, sort_keys=True)
    for k, v in fit.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] > 0:
            busy = v["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (1024 * v["GRBM_GUI_ACTIVE"]["sum"] / 8)
            print("fit kernel %-40s MFMA busy %.3f" % (k[:40], busy))
        if "WRITE_SIZE" in v and "build_train" in k:
            w = v["WRITE_SIZE"]
            print("K(X,X) build: %.1f MB written per launch, %.2f TB/s" % (w["sum"] / w["dispatches"] * 1024 / 1e6, w["sum"] * 1024 / w["total_ns"] / 1e3))
