"""Kernel-level view of one small f_df call (what the multi-start refinement repeats hundreds of times):
rocprofv3 --kernel-trace --stats -d gpurun_out/lat -- python3 tools/latency_prof.py [N] [n_points]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d, m, S = 6, 4, 256
p = R.synthetic_problem(N, d, m, 400, S, 1237)
kern = [B.kern.RBF(d, variance=1.0, lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
model.updateModel(p["X"], p["Y"])
theta = np.array([[0.2 * (j + 1) for j in range(m)]])
U = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), device="neg_sq_dist")
acq = B.uEI_noiseless(model, None, utility=U)
acq.W_samples = p["W"]
x = p["Xc"][:npts]
for _ in range(20):
    acq.acquisition_function_withGradients(x)
t0 = time.perf_counter()
for _ in range(200):
    acq.acquisition_function_withGradients(x)
print("N=%d: f_df(%d points) %.3f ms per call" % (N, npts, (time.perf_counter() - t0) / 200 * 1e3))
