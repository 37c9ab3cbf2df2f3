"""TEST INFRASTRUCTURE ONLY -- extended-precision truth for the parity gates (never imported by the product).

`posterior_truth` runs oracle/truth_ld.c (x87 long double, eps 5.4e-20: kernel, Ky, Cholesky, solves and sums all carried in
long double from the fp64 inputs) and returns what the reference's posterior.py:299-313 / exact_gaussian_inference.py:44-53
define, rounded to double.  It exists so that at cond(Ky) ~ 4e9 (BASELINE configs[2]) the device and the fp64 NumPy oracle are
BOTH measured against something three orders closer to the exact posterior than either:  a parity gate then reads
    err_device <= max(1e-5 * scale, 4 * err_oracle)
instead of comparing two fp64 roundings with each other.

`gate(dev, orc, tru, scale)` evaluates that inequality and returns both errors for printing."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "truth_ld.c")
LIB = os.path.join(HERE, "lib", "libtruth_ld.so")
KIND_ID = {"rbf": 0, "se": 1, "matern52": 2, "matern32": 3}
_lib = None


def build(force=False):
    """gcc -O2 -fopenmp -shared: seconds.  Rebuilt when the source is newer than the library."""
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"])
    return LIB


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.truth_posterior.restype = ctypes.c_int
        _lib.truth_posterior.argtypes = [ctypes.c_int, dp, ctypes.c_int, ctypes.c_int, dp, ctypes.c_double, dp, ctypes.c_double, dp,
                                         ctypes.c_int, dp, dp, dp, dp, dp]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def posterior_truth(kind, X, y, variance, lengthscale, noise, Xc):
    """One output.  Returns dict(mean (C), raw_var (C), mu_train (N), alpha (N), lml): the posterior mean incl. the target
    mean (gp.py:393-399), the RAW variance Kxx - ||L^-1 k*||^2 (posterior.py:309-313: no noise, no clip), the posterior mean at
    the training inputs (multi_outputGP.py:176-180), alpha and the log-marginal (exact_gaussian_inference.py:51-53)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    N, d = X.shape
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(N))
    ls = np.ascontiguousarray(np.broadcast_to(np.asarray(lengthscale, dtype=np.float64).reshape(-1), (d,)))
    Xc = np.ascontiguousarray(np.atleast_2d(Xc), dtype=np.float64)
    C = Xc.shape[0]
    mean, var, mu_train, alpha, lml = np.empty(C), np.empty(C), np.empty(N), np.empty(N), np.empty(1)
    rc = _load().truth_posterior(KIND_ID[kind], _p(X), N, d, _p(y), float(variance), _p(ls), float(noise), _p(Xc), C, _p(mean), _p(var),
                                 _p(mu_train), _p(alpha), _p(lml))
    if rc != 0:
        raise np.linalg.LinAlgError("truth_posterior: info = %d" % rc)
    return dict(mean=mean, raw_var=var, mu_train=mu_train, alpha=alpha, lml=float(lml[0]))


def model_truth(kind, X, Ys, variances, lengthscales, noises, Xc):
    """All m outputs: mean (m, C), var (m, C) = raw + noise clipped at 1e-10 (what model.predict / posterior_variance return,
    gp.py:316-320, gpmodel.py:147), raw_var (m, C), mu_train (m, N)."""
    outs = [posterior_truth(kind, X, Ys[j], variances[j], lengthscales[j], noises[j], Xc) for j in range(len(Ys))]
    raw = np.stack([o["raw_var"] for o in outs])
    noise = np.asarray(noises, dtype=float)[:, None]
    return dict(mean=np.stack([o["mean"] for o in outs]), raw_var=raw, var=np.clip(raw + noise, 1e-10, np.inf),
                mu_train=np.stack([o["mu_train"] for o in outs]), lml=np.array([o["lml"] for o in outs]))


def gate(dev, orc, tru, scale, rel=1e-5, factor=4.0):
    """(ok, err_device, err_oracle, bound): err_x = max |x - truth|; ok = err_device <= max(rel * scale, factor * err_oracle)."""
    dev, orc, tru = (np.asarray(a, dtype=float) for a in (dev, orc, tru))
    e_dev = float(np.abs(dev - tru).max())
    e_orc = float(np.abs(orc - tru).max())
    bound = max(rel * float(scale), factor * e_orc)
    return e_dev <= bound, e_dev, e_orc, bound
