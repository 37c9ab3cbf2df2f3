"""A/B the variance-GEMM options in ONE process (interleaved rounds): python tools/ab_gemm.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R

N, d, m, C, S = 4096, 8, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 1024
p = R.synthetic_problem(N, d, m, C, S, 1237)
kern = [B.kern.RBF(d, variance=1.0, lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
model.incremental = False
t0 = time.perf_counter(); model.updateModel(p["X"], p["Y"]); print("first fit %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for _ in range(3):
    t0 = time.perf_counter(); model.updateModel(p["X"], p["Y"]); print("fit %.2f ms" % ((time.perf_counter() - t0) * 1e3))
model._set_candidates(p["Xc"]); model.set_mc_samples(p["W"])
theta = np.array([[0.2 * (j + 1) for j in range(m)]])
lib = B._ffi.load(); h = model._context().handle
def run(n=3):
    lib.bocf_profile_read(h, None, None, None, 1)
    t0 = time.perf_counter()
    for _ in range(n):
        model._acq_mc_resident(0, 1, None, theta, np.ones(1), None, fetch=False)
    wall = (time.perf_counter() - t0) / n * 1e3
    ms, k, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
    lib.bocf_profile_read(h, ctypes.byref(ms), ctypes.byref(k), ctypes.byref(fl), 1)
    return wall, ms.value / k.value, fl.value / k.value / (ms.value / k.value * 1e-3) / 1e12
model.set_option("profile", 1)
run(1)
variants = [("swizzle", 0), ("swizzle", 256)] if len(sys.argv) <= 2 else [tuple([kv.split("=")[0], int(kv.split("=")[1])]) for kv in sys.argv[2:]]
ref = None
for name, val in variants:                          # results must not depend on the variant
    model.set_option(name, val)
    model._acq_mc_resident(0, 1, None, theta, np.ones(1), None, fetch=False)
    var = model.posterior_variance(p["Xc"][:4096])
    if ref is None:
        ref = var
    else:
        print("%s=%d vs first variant: max |dvar| %.3e, identical %s" % (name, val, np.abs(var - ref).max(), np.array_equal(var, ref)))
model._set_candidates(p["Xc"])
for rnd in range(4):
    for name, val in variants:
        model.set_option(name, val)
        w, g, tf = run()
        print("round %d %s=%d  step %.2f ms  gemm-launch %.2f ms  %.2f TFLOP/s" % (rnd, name, val, w, g, tf))
