"""ctypes binding of libbocf_hip.so (include/bocf_hip.h).  There is no CPU fallback: if the
library is missing or a call fails, an exception is raised.

Two builds of the same sources exist: the PRODUCT library libbocf_hip.so, and libbocf_hip_probes.so (-DBOCF_PROBES) which
additionally holds the timing-only kernel variants and the test hooks (include/bocf_hip.h, bocf_option_info kind 2).  The
product library is what `load()` returns; tools and the tests that need a hook switch with `with probes_library():` (or the
environment variable BOCF_PROBES=1 for a whole process) -- models must be created AND used inside that block."""
import contextlib
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libbocf_hip.so")
PROBES_LIB_PATH = os.path.join(HERE, "lib", "libbocf_hip_probes.so")

KERN_RBF, KERN_SE, KERN_MATERN52, KERN_MATERN32 = 0, 1, 2, 3
ADD_NOISE, CLIP = 1, 2
ACQ_EI, ACQ_PI = 0, 1
UTIL_LINEAR, UTIL_NEG_SQ_DIST, UTIL_NEG_SUM_EXP, UTIL_NEG_EXP_COS, UTIL_ROSENBROCK = 0, 1, 2, 3, 4

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_ll_p = ctypes.POINTER(ctypes.c_longlong)
_ctx_p = ctypes.c_void_p

# bocf_fdf_callback of include/bocf_hip.h
FDF_CALLBACK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, _c_double_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p)

# name -> (restype, argtypes); every symbol include/bocf_hip.h declares
SIGNATURES = {
    "bocf_version": (ctypes.c_int, []),
    "bocf_last_error": (ctypes.c_char_p, []),
    "bocf_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_ctx_p)]),
    "bocf_destroy": (None, [_ctx_p]),
    "bocf_set_option": (ctypes.c_int, [_ctx_p, ctypes.c_char_p, ctypes.c_longlong]),
    "bocf_fit": (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                _c_double_p, _c_double_p, _c_double_p, ctypes.c_int, _c_double_p, _c_double_p]),
    "bocf_set_kernel_ids": (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "bocf_update_targets": (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p]),
    "bocf_append": (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p, _c_double_p]),
    "bocf_lml_gradients": (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p, _c_double_p]),
    "bocf_infer": (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p,
                                  _c_double_p, _c_double_p, ctypes.c_int, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "bocf_last_fit_info": (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "bocf_hmc": (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.c_int,
                                ctypes.POINTER(ctypes.c_int), ctypes.c_double, ctypes.c_double, _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_int,
                                ctypes.c_double, ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                ctypes.POINTER(ctypes.c_int), _c_ll_p]),
    "bocf_hmc_streamed": (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int), ctypes.c_double, ctypes.c_double, _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_double, _c_double_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                         ctypes.POINTER(ctypes.c_int), _c_ll_p]),
    "bocf_get_factor": (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p, _c_double_p]),
    "bocf_get_train_kernel": (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p]),
    "bocf_set_posterior": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p, _c_double_p]),
    "bocf_set_candidates": (ctypes.c_int, [_ctx_p, _c_double_p, ctypes.c_int]),
    "bocf_predict": (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p, _c_double_p]),
    "bocf_predict_cov_column": (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p]),
    "bocf_predict_gradients": (ctypes.c_int, [_ctx_p, _c_double_p, _c_double_p]),
    "bocf_mean_at_train": (ctypes.c_int, [_ctx_p, _c_double_p]),
    "bocf_acq_linear_grad": (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p, _c_double_p, ctypes.c_int, _c_double_p, _c_double_p]),
    "bocf_acq_mc_grad": (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p, ctypes.c_int, _c_double_p, ctypes.c_int, _c_double_p,
                                        ctypes.c_int, _c_double_p, _c_double_p]),
    "bocf_acq_linear": (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_double_p, _c_double_p, ctypes.c_int, _c_double_p]),
    "bocf_set_mc_samples": (ctypes.c_int, [_ctx_p, _c_double_p, ctypes.c_int]),
    "bocf_acq_mc": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.c_int, _c_double_p, ctypes.c_int,
                                   _c_double_p, ctypes.c_int, _c_double_p]),
    "bocf_select_topk": (ctypes.c_int, [_ctx_p, ctypes.c_int, _c_ll_p, _c_double_p]),
    "bocf_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "bocf_comm_init": (ctypes.c_int, [_ctx_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    "bocf_comm_destroy": (ctypes.c_int, [_ctx_p]),
    "bocf_comm_info": (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "bocf_global_topk": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_longlong, _c_ll_p, _c_double_p]),
    "bocf_topk_packed": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bocf_merge_packed": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, _c_ll_p, _c_double_p]),
    "bocf_profile_read": (ctypes.c_int, [_ctx_p, _c_double_p, _c_ll_p, _c_double_p, ctypes.c_int]),
    "bocf_profile_phase": (ctypes.c_int, [_ctx_p, ctypes.c_char_p, _c_double_p, _c_ll_p, ctypes.c_int]),
    "bocf_sync": (ctypes.c_int, [_ctx_p]),
    "bocf_lbfgsb_batched": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_double_p, ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int, _c_double_p, _c_double_p,
                                           _c_ll_p, ctypes.POINTER(ctypes.c_int)]),
    "bocf_get_stat": (ctypes.c_int, [_ctx_p, ctypes.c_char_p, _c_ll_p]),
    "bocf_option_count": (ctypes.c_int, []),
    "bocf_option_info": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), _c_ll_p, _c_ll_p, ctypes.POINTER(ctypes.c_int),
                                        ctypes.POINTER(ctypes.c_char_p)]),
    "bocf_option_check": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_longlong]),
}

_libs = {}
_current = "probes" if os.environ.get("BOCF_PROBES") == "1" else "product"


class BocfHipError(RuntimeError):
    pass


def _load(which):
    if which in _libs:
        return _libs[which]
    path = LIB_PATH if which == "product" else PROBES_LIB_PATH
    if not os.path.exists(path):
        raise ImportError("%s not found at %s -- build it with `python -m bocf_amd.build%s` "
                          "(there is no CPU fallback)" % (os.path.basename(path), path, "" if which == "product" else " --probes"))
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _libs[which] = lib
    return lib


def load():
    """Load the shared library (the product build unless a probes_library() block / BOCF_PROBES=1 is active) and declare its
    signatures.  Raises ImportError when it has not been built (python -m bocf_amd.build)."""
    return _load(_current)


@contextlib.contextmanager
def probes_library():
    """Inside the block `load()` is the -DBOCF_PROBES build (test hooks, timing-only kernel variants)."""
    global _current
    prev, _current = _current, "probes"
    try:
        yield load()
    finally:
        _current = prev


def options(lib=None):
    """The option table of the library: list of (name, lo, hi, kind, description); no GPU needed."""
    lib = lib or load()
    out = []
    for i in range(lib.bocf_option_count()):
        name, what = ctypes.c_char_p(), ctypes.c_char_p()
        lo, hi, kind = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_int()
        lib.bocf_option_info(i, ctypes.byref(name), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(kind), ctypes.byref(what))
        out.append((name.value.decode(), lo.value, hi.value, kind.value, what.value.decode()))
    return out


def dptr(a):
    return None if a is None else a.ctypes.data_as(_c_double_p)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def check(rc, what):
    if rc < 0:
        raise BocfHipError("%s failed (%d): %s" % (what, rc, load().bocf_last_error().decode("utf-8", "replace")))
    return rc


class Context(object):
    """Owns one bocf_ctx*.  Not picklable by design (device handle)."""

    def __init__(self, device=0):
        lib = load()
        h = _ctx_p()
        check(lib.bocf_create(int(device), ctypes.byref(h)), "bocf_create")
        self._h, self._lib, self.device = h, lib, int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bocf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise BocfHipError("context is closed")
        return self._h

    def set_option(self, name, value):
        check(self._lib.bocf_set_option(self.handle, name.encode(), int(value)), "bocf_set_option")

    def stat(self, name):
        v = ctypes.c_longlong()
        check(self._lib.bocf_get_stat(self.handle, name.encode(), ctypes.byref(v)), "bocf_get_stat")
        return v.value
