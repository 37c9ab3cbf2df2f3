"""Variance-contraction tilings at BASELINE configs[1] (N = 1024, m = 4, C = 8192): time of the batch uEI call per option "swizzle"."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bocf_amd as B
from bocf_amd import synthetic as R
N, d, m, C, S = 1024, 6, 4, 8192, 256
p = R.synthetic_problem(N, d, m, C, S, 1236)
kern = [B.kern.RBF(d, variance=1.0, lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
model.updateModel(p["X"], p["Y"])
theta = np.array([[0.2 * (j + 1) for j in range(m)]])
U = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), device="neg_sq_dist")
acq = B.uEI_noiseless(model, None, utility=U)
acq.W_samples = p["W"]
for sw in (258, -1, 0, 258, -1, 0, 256):
    model.set_option("swizzle", sw)
    for _ in range(5): acq._compute_acq(p["Xc"])
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); acq._compute_acq(p["Xc"]); ts.append(time.perf_counter() - t0)
    print("swizzle %4d: %.3f ms per call (median of 30)" % (sw, np.median(ts) * 1e3))
