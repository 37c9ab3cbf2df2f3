"""One Bayesian-optimisation step of a composite objective, entirely on one MI355X, with the reference's
calling pattern (cbo.py:381-405 -> acquisition_optimizer.py:95-154): fit the multi-output GP, score a random
candidate batch with uEI, keep the 16 best anchors, refine them together on batched f_df passes, return the winner.

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
    space = B.Design_space(bounds=[(0.0, 1.0)] * d)
    optimizer = B.AcquisitionOptimizer(space, n_starting=C, n_anchor=16)       # acquisition_optimizer.py:32 (400 there)
    acq = B.uEI_noiseless(model, space, optimizer=optimizer, utility=U)
    acq.W_samples = np.random.normal(size=(256, m))

    model.updateModel(X, Y)                                   # first call allocates the device buffers
    acq.optimize()
    t0 = time.perf_counter()
    model.incremental = False
    model.updateModel(X, Y)
    t_fit = time.perf_counter() - t0
    t0 = time.perf_counter()
    x_best, f_best = acq.optimize()                           # base.py:58-66 -> AcquisitionOptimizer.optimize
    t_opt = time.perf_counter() - t0
    info = optimizer.last_info
    print("fit %.1f ms | acquisition.optimize(): %d random starts scored + top-16 on the device + %d batched f_df passes "
          "(%d points) refining all anchors together: %.1f ms" % (t_fit * 1e3, C, info["f_df_calls"], info["points_evaluated"], t_opt * 1e3))
    print("best anchor score %.6f -> refined %.6f at x = %s" % (info["anchor_points_values"][0], f_best[0, 0], np.round(x_best[0], 4)))
    return x_best, f_best


if __name__ == "__main__":
    main()
