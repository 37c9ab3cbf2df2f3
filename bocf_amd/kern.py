"""Kernel specifications with the constructor signature of the reference's GPy kernels
(GPy/kern/src/stationary.py:59-80, se.py:14-35): plain value holders -- all kernel arithmetic
happens on the device."""
import numpy as np

from . import _ffi


class _Stationary(object):
    kernel_id = None

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name=None):
        self.input_dim = int(input_dim)
        self.ARD = bool(ARD)
        if not ARD:
            lengthscale = np.ones(1) if lengthscale is None else np.asarray(lengthscale, dtype=float).reshape(-1)
            assert lengthscale.size == 1, "Only 1 lengthscale needed for non-ARD kernel"
        else:
            if lengthscale is not None:
                lengthscale = np.asarray(lengthscale, dtype=float).reshape(-1)
                assert lengthscale.size in [1, self.input_dim], "Bad number of lengthscales"
                if lengthscale.size != self.input_dim:
                    lengthscale = np.ones(self.input_dim) * lengthscale
            else:
                lengthscale = np.ones(self.input_dim)
        self.lengthscale = lengthscale
        self.variance = np.atleast_1d(np.asarray(variance, dtype=float))
        assert self.variance.size == 1
        self.name = name or type(self).__name__

    def copy(self):
        return type(self)(self.input_dim, float(self.variance[0]), self.lengthscale.copy(), self.ARD)


class RBF(_Stationary):
    kernel_id = _ffi.KERN_RBF


class SE(_Stationary):
    kernel_id = _ffi.KERN_SE


class Matern52(_Stationary):
    kernel_id = _ffi.KERN_MATERN52


class Matern32(_Stationary):
    kernel_id = _ffi.KERN_MATERN32


_BY_NAME = {"RBF": _ffi.KERN_RBF, "rbf": _ffi.KERN_RBF, "ExpQuad": _ffi.KERN_RBF, "SE": _ffi.KERN_SE, "se": _ffi.KERN_SE,
            "Matern52": _ffi.KERN_MATERN52, "Mat52": _ffi.KERN_MATERN52, "Matern32": _ffi.KERN_MATERN32, "Mat32": _ffi.KERN_MATERN32}


def kernel_spec(kern, input_dim):
    """(kernel_id, variance, lengthscale[d]) of one of our kernels or of a duck-typed GPy kernel
    (class name + .variance + .lengthscale)."""
    kid = getattr(kern, "kernel_id", None)
    if kid is None:
        kid = _BY_NAME.get(type(kern).__name__, _BY_NAME.get(getattr(kern, "name", ""), None))
    if kid is None:
        raise NotImplementedError("kernel %r is not supported on the device (RBF, SE, Matern52, Matern32)" % type(kern).__name__)
    ls = np.asarray(kern.lengthscale, dtype=float).reshape(-1)
    if ls.size == 1:
        ls = np.full(input_dim, float(ls[0]))
    if ls.size != input_dim:
        raise ValueError("kernel has %d lengthscales, data has %d input dimensions" % (ls.size, input_dim))
    return kid, float(np.asarray(kern.variance, dtype=float).reshape(-1)[0]), ls
