"""CPU comparator for tools/hyper_update.py (not a pytest module): the oracle's sequential restatement of
GPModel.updateModel (gpmodel.py:115-120) for ONE output on a bounded sample -- the optimiser run plus `draws` HMC draws of
20 leapfrog steps -- extrapolated to the full update (200 draws per output, m outputs).
    python tests/cpu_baseline_hyper.py [N] [d] [m] [draws]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bocf_amd.synthetic import synthetic_problem
from oracle import cpu_ref as R


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    draws = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    p = synthetic_problem(N, d, m, 8, 8, 1240)
    ref = R.GPHyperRef("se", p["X"], p["Y"][0], 1.0, np.ones(d), 1e-6, True)
    t0 = time.perf_counter()
    R.optimize_hyper(ref, 200)
    t_opt = time.perf_counter() - t0
    n_opt = ref.n_inferences
    np.random.seed(1)
    t0 = time.perf_counter()
    R.hmc_sample(ref, draws, 20, 0.1)
    t_hmc = time.perf_counter() - t0
    per_inf = t_hmc / max(ref.n_inferences - n_opt, 1)
    est = m * (t_opt + per_inf * 200 * 20)
    print("CPU  oracle (NumPy/SciPy, %d threads) N=%d d=%d, one output: optimiser %d inferences %.2f s; HMC %.2f ms per inference "
          "(%d draws sampled) => %.1f s for the full update of %d outputs (200 draws x 20 leapfrog steps each)"
          % (os.cpu_count(), N, d, n_opt, t_opt, 1e3 * per_inf, draws, est, m))


if __name__ == "__main__":
    main()
