"""Latency of single-point calls (what L-BFGS refinement does): python tools/latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R
N, d, m = 4096, 8, 4
p = R.synthetic_problem(N, d, m, 400, 25, 1237)
kern = [B.kern.RBF(d, variance=1.0, lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
model.updateModel(p["X"], p["Y"])
theta = np.array([[0.2 * (j + 1) for j in range(m)]])
U = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), device="neg_sq_dist")
acq = B.uEI_noiseless(model, None, utility=U)
mae = B.maEI(model, None, utility=B.Utility(parameter_dist=B.ParameterDistribution(support=theta / 2, prob_dist=np.ones(1)), linear=True))
def t(f, n=50):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
x1 = p["Xc"][:1]
for sp in (1, 0):
    model.set_option("small_path", sp)
    print("small_path=%d  predict(1) %.3f ms | uEI f(1) %.3f | uEI f_df(1) %.3f | maEI f_df(1) %.3f | uEI f(400) %.3f | uEI f_df(16) %.3f" % (
        sp, t(lambda: model.predict(x1)), t(lambda: acq.acquisition_function(x1)), t(lambda: acq.acquisition_function_withGradients(x1)),
        t(lambda: mae.acquisition_function_withGradients(x1)), t(lambda: acq.acquisition_function(p["Xc"]), 20),
        t(lambda: acq.acquisition_function_withGradients(p["Xc"][:16]))))
big = R.synthetic_problem(64, d, m, 65536, 1024, 1237)
acq.W_samples = big["W"]
model.set_option("small_path", 1)
print("PCIe-inclusive uEI acquisition_function(65536 x S=1024) through the Python surface: %.2f ms" % t(lambda: acq.acquisition_function(big["Xc"]), 5))
model._set_candidates(big["Xc"]); model.set_mc_samples(big["W"])
print("resident: %.2f ms" % t(lambda: model._acq_mc_resident(0, 1, None, theta, np.ones(1), None, fetch=False), 5))
