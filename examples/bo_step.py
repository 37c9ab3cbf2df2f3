"""One Bayesian-optimisation step of a composite objective, entirely on one MI355X, with the reference's
calling pattern (cbo.py:381-405 -> acquisition_optimizer.py:95-154): fit the multi-output GP, score a random
candidate batch with uEI, keep the 16 best anchors, refine them with L-BFGS-B on f_df, return the winner.

    python examples/bo_step.py [N] [C]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B


def objective(X):                       # m = 3 outputs of a toy simulator
    return [np.sin(3 * X[:, :1]) + X[:, 1:2] ** 2, np.cos(2 * X[:, 1:2]) * X[:, :1], (X ** 2).sum(1, keepdims=True)]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    d, m = 2, 3
    np.random.seed(1)
    X = np.random.uniform(size=(N, d))
    Y = objective(X)
    target = np.array([[0.8, 0.1, 0.4]])                     # utility: -|| f(x) - target ||^2  (test_1a.py:89-92)
    dist = B.ParameterDistribution(continuous=False, support=target, prob_dist=np.ones(1))
    U = B.Utility(parameter_dist=dist, device="neg_sq_dist")
    model = B.multi_outputGP(m, kernel=[B.kern.SE(d, variance=1., lengthscale=0.4)] * m, noise_var=[1e-6] * m, fixed_hyps=True)
    acq = B.uEI_noiseless(model, None, utility=U)
    acq.W_samples = np.random.normal(size=(256, m))

    model.updateModel(X, Y)                                   # first call allocates the device buffers
    acq.acquisition_function_withGradients(X[:16])
    t0 = time.perf_counter()
    model.incremental = False
    model.updateModel(X, Y)
    t_fit = time.perf_counter() - t0
    # candidate batch drawn like samples_multidimensional_uniform (random_design.py:67-77)
    Xc = np.empty((C, d))
    for k in range(d):
        Xc[:, k] = np.random.uniform(low=0.0, high=1.0, size=C)
    t0 = time.perf_counter()
    scores = acq.acquisition_function(Xc)                     # -acq, (C, 1): one device pass
    anchors = Xc[acq.select_anchors(16)]                      # np.argsort(scores)[:16], on the device
    t_batch = time.perf_counter() - t0
    import scipy.optimize                                     # the caller's optimiser, as in optimizer.py:334-354
    t0 = time.perf_counter()
    refined, calls = [], [0]

    def f_df(x):
        calls[0] += 1
        f, g = acq.acquisition_function_withGradients(x[None])
        return float(f[0, 0]), g[0]
    for a in anchors:
        res = scipy.optimize.fmin_l_bfgs_b(f_df, x0=a, bounds=[(0.0, 1.0)] * d, maxiter=50, factr=1e5, pgtol=1e-15)
        refined.append((np.atleast_2d(res[0]), np.atleast_2d(res[1])))
    t_ref = time.perf_counter() - t0
    x_best, f_best = min(refined, key=lambda t: t[1][0, 0])
    print("fit %.1f ms | batch of %d candidates %.1f ms | L-BFGS-B refinement of 16 anchors: %d single-point f_df calls, %.1f ms"
          % (t_fit * 1e3, C, t_batch * 1e3, calls[0], t_ref * 1e3))
    print("best anchor score %.6f -> refined %.6f at x = %s" % (scores.min(), f_best[0, 0], np.round(x_best[0], 4)))
    return x_best, f_best


if __name__ == "__main__":
    main()
