"""A few iterations of composite Bayesian optimisation with the reference's loop structure (cbo.py:274-330,381-405):
learn the hyper-parameters of a multi-output GP (optimise + HMC, 10 hyper-samples), optimise uEI over them
(random starts scored on the device -> top-16 anchors -> all anchors refined together), evaluate the objective at the
suggestion, append, repeat -- every GP inference, prediction and acquisition value computed on the MI355X.

    python examples/bo_loop.py [iterations] [n_starting] [--quick]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B


def simulator(X):                        # m = 2 attributes of a toy simulator on [0, 1]^2
    X = np.atleast_2d(X)
    return [np.sin(3 * X[:, :1]) * X[:, 1:2] + X[:, :1] ** 2, np.cos(2 * X[:, 1:2]) + 0.5 * X[:, :1]]


def run(iterations=5, n_starting=4096, quick=False, seed=1, verbose=True):
    d, m = 2, 2
    np.random.seed(seed)
    space = B.Design_space([{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': d}])
    X = np.random.uniform(size=(2 * (d + 1), d))                               # initial design (test_2a.py:74)
    Y = simulator(X)
    target = np.array([[0.9, 0.6]])                                            # utility: -|| f(x) - target ||^2
    U = B.Utility(parameter_dist=B.ParameterDistribution(support=target, prob_dist=np.ones(1)), device="neg_sq_dist")
    model = B.multi_outputGP(output_dim=m, exact_feval=[True] * m, fixed_hyps=False)     # test_2a.py:49
    if quick:                                                                  # shorter chains than gpmodel.py:32 (smoke runs)
        model.n_burnin, model.subsample_interval, model.leapfrog_steps = 20, 2, 5
    optimizer = B.AcquisitionOptimizer(space, optimizer='lbfgs', n_starting=n_starting, n_anchor=16)
    acq = B.uEI_noiseless(model, space, optimizer=optimizer, utility=U)
    history = []
    for it in range(iterations):
        t0 = time.perf_counter()
        model.updateModel(X, Y)                                                # cbo.py:394: optimise + HMC + 10 hyper-samples
        t_model = time.perf_counter() - t0
        t0 = time.perf_counter()
        x_next, acq_val = acq.optimize()                                       # sequential.py:22 -> base.py:58-66
        t_acq = time.perf_counter() - t0
        try:                                                                   # cbo.py:299-302, verbatim: the call omits the
            acq.update_Z_samples()                                             # required argument, the TypeError is swallowed
        except Exception:                                                      # and W_samples is never redrawn
            pass
        y_next = simulator(x_next)
        X = np.vstack((X, x_next))                                             # cbo.py:304
        Y = [np.vstack((Y[j], y_next[j])) for j in range(m)]
        util = -np.sum((np.hstack(Y) - target) ** 2, axis=1)
        history.append(float(util.max()))
        if verbose:
            print("iteration %d: model update %.2f s (%d inferences), acquisition optimisation %.1f ms (%d f_df passes) -> x = %s, "
                  "acq = %.3g, best utility so far %.5f" % (it + 1, t_model, model.last_update_info["hmc_inferences"] +
                                                            model.last_update_info["optimizer_inferences"], 1e3 * t_acq,
                                                            optimizer.last_info["f_df_calls"], np.round(x_next[0], 4), -float(np.ravel(acq_val)[0]),
                                                            history[-1]))
    return X, Y, history


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    run(int(args[0]) if args else 5, int(args[1]) if len(args) > 1 else 4096, quick="--quick" in sys.argv)
