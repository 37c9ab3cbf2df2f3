"""updateModel latency: full fit vs rank-1 append vs targets-only refresh (N ~ 4000, m = 4)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R
N, d, m = 4000, 8, 4
p = R.synthetic_problem(N + 40, d, m, 64, 8, 1237)
kern = [B.kern.RBF(d, variance=1.0, lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
X, Ys = p["X"], p["Y"]
model.updateModel(X[:N], [y[:N] for y in Ys])
t0 = time.perf_counter(); model.incremental = False; model.updateModel(X[:N], [y[:N] for y in Ys]); t_fit = time.perf_counter() - t0
model.incremental = True
ts = []
for n in range(N + 1, N + 31):
    t0 = time.perf_counter(); model.updateModel(X[:n], [y[:n] for y in Ys]); ts.append(time.perf_counter() - t0)
t0 = time.perf_counter(); model.updateModel(X[:N + 30], [y[:N + 30] * 1.1 for y in Ys]); t_y = time.perf_counter() - t0
print("N=%d m=%d: full fit %.2f ms | append median %.3f ms | targets-only %.3f ms" % (N, m, t_fit * 1e3, np.median(ts) * 1e3, t_y * 1e3))
# (accuracy after appends: tests/test_gpu_parity.py::test_incremental_update_*)
