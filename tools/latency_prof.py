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
x1 = p["Xc"][:1]
for _ in range(200):
    acq.acquisition_function_withGradients(x1)
