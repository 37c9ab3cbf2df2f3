"""Time one acquisition-optimisation step on the GPU: score n_starting random starts, device top-16,
refine all anchors together (bocf_amd.AcquisitionOptimizer) -- beside the reference's schedule on the same device
functions (one scipy L-BFGS-B per anchor, single-point f_df calls).   python tools/bo_iteration.py [N] [n_starting] [d] [m] [S]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import acquisition_optimizer as AO
from bocf_amd import synthetic as R


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    m = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    S = int(sys.argv[5]) if len(sys.argv) > 5 else 256
    p = R.synthetic_problem(N, d, m, 8, S, 1236, noise=1e-4)
    model = B.multi_outputGP(m, kernel=[B.kern.RBF(d, variance=1.0, lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)],
                             noise_var=p["noise"], fixed_hyps=True)
    model.updateModel(p["X"], p["Y"])
    theta = np.array([[0.5] * m])
    U = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), device="neg_sq_dist")
    space = AO.Design_space(bounds=[(0.0, 1.0)] * d)
    import scipy.optimize
    for name in ("uEI", "maEI"):
        if name == "uEI":
            acq = B.uEI_noiseless(model, space, optimizer=AO.AcquisitionOptimizer(space, n_starting=C, n_anchor=16), utility=U)
            acq.W_samples = p["W"]
        else:
            Ul = B.Utility(parameter_dist=B.ParameterDistribution(support=np.full((1, m), 1.0 / m), prob_dist=np.ones(1)), linear=True)
            acq = B.maEI(model, space, optimizer=AO.AcquisitionOptimizer(space, n_starting=C, n_anchor=16), utility=Ul)
        np.random.seed(3)
        acq.optimize()                       # warm-up (allocations)
        ts = []
        for rep in range(5):
            np.random.seed(3)
            t0 = time.perf_counter()
            x, fx = acq.optimize()
            ts.append(time.perf_counter() - t0)
        info = acq.optimizer.last_info
        # the reference's schedule on the same device functions
        np.random.seed(3)
        t0 = time.perf_counter()
        X0 = AO.samples_multidimensional_uniform(space.get_bounds(), C)
        sc = acq.acquisition_function(X0).flatten()
        anchors = X0[np.argsort(sc)[:16]]
        t_score = time.perf_counter() - t0
        calls = [0]

        def f_df(z):
            calls[0] += 1
            f, g = acq.acquisition_function_withGradients(z[None])
            return float(f[0, 0]), g[0]
        t0 = time.perf_counter()
        outs = []
        for a in anchors:
            res = scipy.optimize.fmin_l_bfgs_b(f_df, x0=a, bounds=space.get_bounds(), maxiter=500, factr=1e6)
            outs.append(float(acq.acquisition_function(np.atleast_2d(res[0]))[0, 0]))
        t_seq = time.perf_counter() - t0
        print("%s N=%d d=%d m=%d S=%d n_starting=%d: batched optimize() %.1f ms (min of 5; %d batched f_df passes, %d points, max %d iterations) -> f=%.8g"
              % (name, N, d, m, S, C, min(ts) * 1e3, info["f_df_calls"], info["points_evaluated"], info["iterations"].max(), fx[0, 0]))
        print("    reference schedule on the device functions: scoring+argsort %.1f ms, 16 sequential L-BFGS-B %.1f ms (%d single-point f_df calls) -> f=%.8g"
              % (t_score * 1e3, t_seq * 1e3, calls[0], min(outs)))


if __name__ == "__main__":
    main()
