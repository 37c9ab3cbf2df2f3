# Timing-only variants of the diagonal-block kernel (option potrf_scalar = 11..14, results wrong): kernel duration from a kernel trace
cd /tmp && export TMPDIR=/tmp
for v in ${1:-0 2 11}; do
  rm -rf /tmp/pp; BOCF_OPTIONS=potrf_scalar=$v,lookahead=0,overlap_inverse=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py 2048 4 > /tmp/pp.log 2>&1
  python3 - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/pp/*/*kernel_stats.csv")[0])):
    if "potrf_diag" in r["Name"]:
        print("variant $v: %s calls %s avg %.1f us min %.1f us" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3), flush=True)
PY
done
