"""Fit-only loop for profiling: python tools/fit_only.py [N] [m]   (BOCF_OPTIONS=name=value,... passes options through)
Prints the wall time of five fits and the HIP-event time of the fit's phases (option "profile")."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = 8
p = R.synthetic_problem(N, d, m, 128, 8, 1237)
kern = [B.kern.RBF(d, variance=1.0, lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
model.incremental = False
for kv in os.environ.get("BOCF_OPTIONS", "").split(","):      # e.g. BOCF_OPTIONS=lookahead=0
    if kv:
        model.set_option(kv.split("=")[0], int(kv.split("=")[1]))
model.updateModel(p["X"], p["Y"])
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    model.updateModel(p["X"], p["Y"])
    ts.append((time.perf_counter() - t0) * 1e3)
lib, h = B._ffi.load(), model._context().handle
model.set_option("profile", 1)
reps = 3
for _ in range(reps):
    model.updateModel(p["X"], p["Y"])
out = []
for name in ("kbuild", "cholesky", "inverse", "alpha"):
    ms, n = ctypes.c_double(), ctypes.c_longlong()
    lib.bocf_profile_phase(h, name.encode(), ctypes.byref(ms), ctypes.byref(n), 1)
    out.append("%s %.3f" % (name, ms.value / reps))
ctx = model._context()
print("N=%d m=%d [%s] fit ms: %s | phases ms: %s | schedule %d, dependency time-outs %d" % (
    N, m, os.environ.get("BOCF_OPTIONS", ""), " ".join("%.2f" % t for t in ts), ", ".join(out), ctx.stat("last_schedule"), ctx.stat("sched_timeouts")))
