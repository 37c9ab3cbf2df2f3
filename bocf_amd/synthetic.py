"""The synthetic workload of SURVEY.md section 8(d) (what bench.py, the tools and the parity tests feed the path):
X ~ U[0,1]^{N x d}; Y_j = sin(2 pi x.a_j) + 0.5 cos(3 x.b_j) with a_j, b_j ~ N(0, I); sigma_f^2 = 1; ARD lengthscales
0.5 sqrt(d) (1 +- 0.2 jitter per output); candidates drawn column by column like samples_multidimensional_uniform
(GPyOpt/experiment_design/random_design.py:67-77); MC normals W = rng.normal(size=(S, m)) drawn on the host as the
reference does (uEI_noiseless.py:31).  Input generation only -- no GP arithmetic."""
import numpy as np


def synthetic_problem(N, d, m, C, S, seed, noise=1e-6):
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(N, d))
    Y = []
    for j in range(m):
        a, b = rng.normal(size=d), rng.normal(size=d)
        Y.append((np.sin(2 * np.pi * X.dot(a)) + 0.5 * np.cos(3 * X.dot(b)))[:, None])
    ls = [0.5 * np.sqrt(d) * (1.0 + 0.2 * rng.uniform(-1, 1, size=d)) for _ in range(m)]
    Xc = np.empty((C, d))
    for k in range(d):
        Xc[:, k] = rng.uniform(low=0.0, high=1.0, size=C)
    W = rng.normal(size=(S, m))
    return dict(X=X, Y=Y, lengthscales=ls, variances=[1.0] * m, noise=[noise] * m, Xc=Xc, W=W)
