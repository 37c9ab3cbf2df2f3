"""Latency of one batched hyper-parameter inference (bocf_fit + bocf_lml_gradients for m outputs): the unit of work of
the HMC / optimiser loops (tools/hyper_update.py).   python tools/infer_latency.py [N] [d] [m] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    p = R.synthetic_problem(N, d, m, 8, 8, 1240)
    model = B.multi_outputGP(m, exact_feval=[True] * m, fixed_hyps=False)
    model._X, model._Y = p["X"], p["Y"]
    model._create_sampler_state()
    params = [o.expanded(d) for o in model._sampler_outputs]
    model._infer(params)
    model.set_option("reuse_data", 1)          # what updateModel does for the inferences after the first
    model.set_option("skip_mu_train", 1)
    t0 = time.perf_counter()
    for i in range(reps):
        params[0] = (1.0 + 1e-3 * (i % 7), params[0][1], params[0][2])
        model._infer(params)
    dt = (time.perf_counter() - t0) / reps
    print("N=%d d=%d m=%d: %.3f ms per batched inference (%d reps)" % (N, d, m, dt * 1e3, reps))


if __name__ == "__main__":
    main()
