"""Distribution of the wall time of repeated fits: python tools/fit_jitter.py N m reps  (looks for stalls of a schedule)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R
N, m, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = R.synthetic_problem(N, 8, m, 128, 8, 1237)
kern = [B.kern.RBF(8, variance=1.0, lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
model.incremental = False
for kv in os.environ.get("BOCF_OPTIONS", "").split(","):
    if kv:
        model.set_option(kv.split("=")[0], int(kv.split("=")[1]))
model.updateModel(p["X"], p["Y"])
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    model.updateModel(p["X"], p["Y"])
    ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print("N=%d m=%d [%s] %d fits: min %.2f median %.2f p90 %.2f max %.2f ms; > 1.5 x median: %d" % (
    N, m, os.environ.get("BOCF_OPTIONS", ""), reps, ts.min(), np.median(ts), np.percentile(ts, 90), ts.max(), int((ts > 1.5 * np.median(ts)).sum())))
