"""Utility / ParameterDistribution / ExpectationUtility with the reference's signatures
(utility.py:6-48, parameter_distribution.py:5-29, expectation_utility.py:3-9), plus the
DEVICE specification of the utility: the Monte-Carlo acquisitions evaluate U on the GPU, so U
must be one of the closed set of utilities the reference's experiment scripts use."""
import numpy as np

from . import _ffi


class ParameterDistribution(object):
    """parameter_distribution.py:5-29 (same attributes, same global-RNG sampling call)."""

    def __init__(self, continuous=False, support=None, prob_dist=None, sample_generator=None):
        if continuous is True and sample_generator is None:
            pass
        else:
            self.continuous = continuous
            self.support = support
            self.prob_dist = prob_dist
            self.sample_generator = sample_generator
        if support is not None and len(support) < 20:
            self.use_full_support = True
        else:
            self.use_full_support = False

    def sample(self, n_samples):
        if self.continuous:
            parameter_samples = self.sample_generator(n_samples)
        else:
            indices = np.random.choice(int(len(self.support)), size=n_samples, p=self.prob_dist)
            parameter_samples = self.support[indices, :]
        return parameter_samples


class ExpectationUtility(object):
    """expectation_utility.py:3-9."""

    def __init__(self, func, gradient):
        self.func = func
        self.gradient = gradient


_DEVICE_KINDS = {"linear": _ffi.UTIL_LINEAR, "neg_sq_dist": _ffi.UTIL_NEG_SQ_DIST, "neg_sum_exp": _ffi.UTIL_NEG_SUM_EXP,
                 "neg_exp_cos": _ffi.UTIL_NEG_EXP_COS, "rosenbrock": _ffi.UTIL_ROSENBROCK}


def _host_func(kind, params):
    """NumPy form of a device utility (for user code that evaluates U on observed data, e.g.
    cbo.py's bookkeeping) -- the acquisitions never call it."""
    if kind == "linear":
        return lambda parameter, y: np.dot(parameter, y)
    if kind == "neg_sq_dist":
        return lambda parameter, y: -np.sum(np.square((np.asarray(y).transpose() - parameter).transpose()), axis=0)
    if kind == "neg_sum_exp":
        return lambda parameter, y: np.sum(-np.exp(y), axis=0)
    if kind == "neg_exp_cos":
        c = np.asarray(params, dtype=float)
        return lambda parameter, y: -np.tensordot(c, np.exp(-np.asarray(y) / np.pi) * np.cos(np.pi * np.asarray(y)), axes=(0, 0))
    if kind == "rosenbrock":
        def f(a, y):
            y = np.asarray(y)
            h = y.shape[0] // 2
            return -(np.sum((np.atleast_1d(a)[0] - y[:h]) ** 2, axis=0) + 100.0 * np.sum(y[h:2 * h] ** 2, axis=0))
        return f
    raise ValueError(kind)


class Utility(object):
    """
    utility.py:6-48 plus `device`: name of the device utility ("linear", "neg_sq_dist",
    "neg_sum_exp", "neg_exp_cos", "rosenbrock") and `device_params` (weights c of neg_exp_cos).
    `func` may be omitted for a device utility.
    """

    def __init__(self, func=None, dfunc=None, parameter_dist=None, linear=False, device=None, device_params=None):
        if device is None and linear:
            device = "linear"
        if device is not None and device not in _DEVICE_KINDS:
            raise ValueError("unknown device utility %r (have: %s)" % (device, ", ".join(sorted(_DEVICE_KINDS))))
        self.device = device
        self.device_params = None if device_params is None else np.asarray(device_params, dtype=float)
        self.func = func if func is not None else (_host_func(device, device_params) if device else None)
        self.dfunc = dfunc
        self.parameter_dist = parameter_dist
        self.linear = linear

    def device_kind(self, m=None):
        """Enum of the device utility.  When only a Python callable was given (the reference's scripts: utility.py:6-14 has
        no `device` argument) the callable is RECOGNISED: it is probed at a handful of fixed points (private RNG, the global
        np.random stream is not touched) and compared with the closed set of device utilities; `m` is the number of model
        outputs (needed when the utility parameter does not determine it).  No match -> NotImplementedError."""
        if self.device is None and self.func is not None:
            self._recognise(m)
        if self.device is None:
            raise NotImplementedError(
                "this Utility wraps a Python callable that is none of the device utilities; the Monte-Carlo acquisitions run on "
                "the GPU and need one of %s (Utility(..., device=...))" % ", ".join(sorted(_DEVICE_KINDS)))
        return _DEVICE_KINDS[self.device]

    def _recognise(self, m):
        support = getattr(self.parameter_dist, "support", None)
        if support is None or len(support) == 0:
            return
        theta = np.asarray(support[0], dtype=float).reshape(-1)
        if m is None:
            m = theta.size
        rng = np.random.RandomState(20180101)
        probes = [rng.uniform(-1.0, 1.0, size=m) for _ in range(3 * m + 4)]

        def user(y):
            return float(np.squeeze(self.func(theta, y)))
        try:
            want = np.array([user(y) for y in probes])
        except Exception:
            return
        if not np.all(np.isfinite(want)):
            return

        def matches(kind, params=None):
            if kind in ("linear", "neg_sq_dist") and theta.size != m:
                return False
            if kind == "rosenbrock" and (m % 2 or theta.size < 1):
                return False
            f = _host_func(kind, params)
            got = np.array([float(np.squeeze(f(theta, y))) for y in probes])
            return bool(np.all(np.abs(got - want) <= 1e-9 * (1.0 + np.abs(want))))
        for kind in ("linear", "neg_sq_dist", "neg_sum_exp", "rosenbrock"):
            if matches(kind):
                self.device = kind
                return
        # -sum_j c_j exp(-y_j / pi) cos(pi y_j): the weights are recovered from the first 3m probes and must reproduce the rest
        G = np.array([np.exp(-y / np.pi) * np.cos(np.pi * y) for y in probes])
        c, *_ = np.linalg.lstsq(-G[:3 * m], want[:3 * m], rcond=None)
        if matches("neg_exp_cos", c):
            self.device, self.device_params = "neg_exp_cos", c

    def evaluate_w_gradient(self, parameter, y):
        return self.eval_func(parameter, y), self.eval_gradient(parameter, y)

    def eval_func(self, parameter, y):
        return self.func(parameter, y)

    def eval_gradient(self, parameter, y):
        return self.dfunc(parameter, y)
