"""Time one multi_outputGP(fixed_hyps=False).updateModel with the reference's sampler settings (gpmodel.py:32,115-120:
<= 200 optimiser steps, 200 HMC draws x 20 leapfrog steps per output) on the GPU.  (The CPU comparator -- the oracle's sequential restatement of the same flow on a
bounded sample -- is tests/cpu_baseline_hyper.py.)   python tools/hyper_update.py [N] [d] [m]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    p = R.synthetic_problem(N, d, m, 8, 8, 1240)
    X, Ys = p["X"], p["Y"]
    model = B.multi_outputGP(m, exact_feval=[True] * m, fixed_hyps=False)          # test_2a.py:49
    model.device_hmc = os.environ.get("BOCF_DEVICE_HMC", "1") != "0"               # 0: lockstep host loop (one device inference per leapfrog step)
    model.device_hmc_streamed = os.environ.get("BOCF_DEVICE_HMC_STREAMED", "1") != "0"
    np.random.seed(0)
    t0 = time.perf_counter()
    model.updateModel(X, Ys)
    t_first = time.perf_counter() - t0
    np.random.seed(1)
    t0 = time.perf_counter()
    model.updateModel(X, Ys)
    t_upd = time.perf_counter() - t0
    info = model.last_update_info
    n_inf = info["optimizer_inferences"] + info["hmc_inferences"]
    print("GPU  N=%d d=%d m=%d [%s]: updateModel %.2f s (first call %.2f s): %d optimiser + %d HMC batched inferences (each = %d fits), "
          "%.3f ms per batched inference; accepted %s of %d, diverged-rejected handled; final fit of %d hyper-samples x %d outputs"
          % (N, d, m, "resident HMC chain" if model.device_hmc and N <= 128 and d <= 16 else ("stream-resident HMC chain" if model.device_hmc and model.device_hmc_streamed else "lockstep host loop"), t_upd, t_first, info["optimizer_inferences"], info["hmc_inferences"], m, 1e3 * t_upd / max(n_inf, 1),
             info["accepted"].tolist(), info["num_samples"], model._H, m))
    Xc = np.random.RandomState(2).uniform(size=(4096, d))
    theta = np.full((1, m), 1.0 / m)
    acq = B.maEI(model, None, utility=B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), linear=True))
    acq._compute_acq(Xc)
    t0 = time.perf_counter()
    acq._compute_acq(Xc)
    print("     maEI over %d candidates averaged over %d hyper-samples: %.2f ms" % (Xc.shape[0], acq.n_hyps_samples, 1e3 * (time.perf_counter() - t0)))


if __name__ == "__main__":
    main()
