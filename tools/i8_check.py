"""predict_i8 against the fp64 contraction: python tools/i8_check.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R
for N, m, C in ((130, 2, 300), (700, 3, 1000), (1024, 4, 8192), (2500, 1, 5000), (4096, 4, 65536)):
    d = 6
    p = R.synthetic_problem(N, d, m, C, 8, 4242 + N, noise=1e-5)
    kern = [B.kern.RBF(d, variance=p["variances"][j], lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
    model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
    model.updateModel(p["X"], p["Y"])
    out = []
    for opt in (0, 1):
        model.set_option("predict_i8", opt)
        model.predict(p["Xc"])
        t0 = time.perf_counter()
        mean, var = model.predict(p["Xc"])
        out.append((mean, var, (time.perf_counter() - t0) * 1e3))
    dv = np.abs(out[1][1] - out[0][1]).max()
    print("N=%d m=%d C=%d: max |var_i8 - var_f64| = %.3g (var range %.3g .. %.3g), mean identical %s; predict() %.2f ms fp64, %.2f ms int8" % (
        N, m, C, dv, out[0][1].min(), out[0][1].max(), np.array_equal(out[0][0], out[1][0]), out[0][2], out[1][2]))
