"""predict() wall time, fp64 against predict_i8, on small models (repeated calls): python tools/i8_small_timing.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R
for N, m, C in ((700, 3, 1000), (600, 3, 2000), (1900, 2, 40000), (768, 1, 1024)):
    d = 6
    p = R.synthetic_problem(N, d, m, C, 8, 4242 + N, noise=1e-5)
    kern = [B.kern.RBF(d, variance=p["variances"][j], lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
    model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
    model.updateModel(p["X"], p["Y"])
    for opt in (0, 1):
        model.set_option("predict_i8", opt)
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            model.predict(p["Xc"])
            ts.append((time.perf_counter() - t0) * 1e3)
        print("N=%d m=%d C=%d predict_i8=%d: %s ms" % (N, m, C, opt, " ".join("%.2f" % t for t in ts)))
