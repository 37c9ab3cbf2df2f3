"""Acquisition plug-ins with the reference's class surface (GPyOpt/acquisitions/base.py:5-74,
maEI.py, maPI.py, EI.py, PI.py, uEI_noiseless.py, uPI.py) whose `_compute_acq` runs on the
MI355X: predict (K* build, N^2 C contraction) + acquisition in one device pass per call.

The host keeps exactly the reference's RNG touchpoints (W_samples drawn with np.random.normal
at construction, theta sampled with np.random.choice) so seeded trajectories line up.
There is no CPU fallback: the model must be a bocf_amd.multi_outputGP.
"""
import warnings

import numpy as np

from . import _ffi


def constant_cost_withGradients(x):
    """GPyOpt/core/task/cost.py:76-80."""
    return np.ones(x.shape[0])[:, None], np.zeros(x.shape)


class AcquisitionBase(object):
    """GPyOpt/acquisitions/base.py:5-74 (fork: acquisition_function returns -acq, cost and
    constraint weighting commented out)."""

    analytical_gradient_prediction = False

    def __init__(self, model, space, optimizer, cost_withGradients=None):
        self.model = model
        self.space = space
        self.optimizer = optimizer
        self.analytical_gradient_acq = self.analytical_gradient_prediction and self.model.analytical_gradient_prediction
        if cost_withGradients is None:
            self.cost_withGradients = constant_cost_withGradients
        else:
            self.cost_withGradients = cost_withGradients

    @staticmethod
    def fromDict(model, space, optimizer, cost_withGradients, config):
        raise NotImplementedError()

    def acquisition_function(self, x):
        f_acqu = self._compute_acq(x)
        return -f_acqu

    def acquisition_function_withGradients(self, x):
        f_acqu, df_acqu = self._compute_acq_withGradients(x)
        return -f_acqu, -df_acqu

    def optimize(self, duplicate_manager=None, x_baseline=None):
        if not self.analytical_gradient_acq:
            out = self.optimizer.optimize(f=self.acquisition_function, duplicate_manager=duplicate_manager, x_baseline=x_baseline)
        else:
            out = self.optimizer.optimize(f=self.acquisition_function, f_df=self.acquisition_function_withGradients,
                                          duplicate_manager=duplicate_manager, x_baseline=x_baseline)
        return out

    def _compute_acq(self, x):
        raise NotImplementedError('')

    def _compute_acq_withGradients(self, x):
        raise NotImplementedError('')

    # -- shared by the subclasses
    def _device_model(self):
        if not hasattr(self.model, "acq_linear"):
            raise TypeError("bocf_amd acquisitions need a bocf_amd.multi_outputGP model (no CPU fallback)")
        return self.model

    def select_anchors(self, num_anchor=16):
        """Indices (into the last evaluated batch) of the `num_anchor` best candidates, computed on
        the device: np.argsort(acquisition_function(X).flatten())[:num_anchor] of
        anchor_points_generator.py:59-61."""
        return self._device_model().select_topk(num_anchor)[0]


class _ClosedForm(AcquisitionBase):
    """maEI.py:9-163 / maPI.py:9-158: closed-form EI / PI of a linear utility theta.f(x),
    averaged over the utility-parameter distribution."""
    analytical_gradient_prediction = True
    _kind = _ffi.ACQ_EI
    _n_theta_samples = 3          # maEI.py:46; maPI.py:46 draws 10

    def __init__(self, model, space, optimizer=None, cost_withGradients=None, utility=None):
        self.optimizer = optimizer
        self.utility = utility
        super(_ClosedForm, self).__init__(model, space, optimizer, cost_withGradients=cost_withGradients)
        if cost_withGradients is not None:
            print('LBC acquisition does now make sense with cost. Cost set to constant.')
        self.cost_withGradients = constant_cost_withGradients
        self.use_full_support = self.utility.parameter_dist.use_full_support
        self.n_hyps_samples = min(10, self.model.number_of_hyps_samples())

    def _compute_acq(self, X):
        if self.use_full_support:
            self.utility_params_samples = self.utility.parameter_dist.support
            self.utility_param_dist = np.atleast_1d(self.utility.parameter_dist.prob_dist)
            prob = self.utility_param_dist
        else:
            self.utility_params_samples = self.utility.parameter_dist.sample(self._n_theta_samples)
            prob = None
        X = np.atleast_2d(X)
        thetas = np.asarray(self.utility_params_samples, dtype=float).reshape(len(self.utility_params_samples), -1)
        # the h-loop of maEI.py:85-98 runs on the device (with fixed hyper-parameters its identical passes are one pass)
        acqX = self._device_model().acq_linear(X, self._kind, thetas, prob, n_hyps=self.n_hyps_samples)
        return np.reshape(acqX, (X.shape[0], 1))

    def _compute_acq_withGradients(self, X):
        """maEI.py:57-78 / maPI.py:56-76: value and d/dX; not-full-support draws 3 thetas (both classes)."""
        if self.use_full_support:
            self.utility_params_samples = self.utility.parameter_dist.support
            self.utility_param_dist = np.atleast_1d(self.utility.parameter_dist.prob_dist)
            prob = self.utility_param_dist
        else:
            self.utility_params_samples = self.utility.parameter_dist.sample(3)
            prob = None
        X = np.atleast_2d(X)
        thetas = np.asarray(self.utility_params_samples, dtype=float).reshape(len(self.utility_params_samples), -1)
        acqX, dacq_dX = self._device_model().acq_linear_grad(X, self._kind, thetas, prob, n_hyps=self.n_hyps_samples)
        return np.reshape(acqX, (X.shape[0], 1)), np.reshape(dacq_dX, X.shape)


class maEI(_ClosedForm):
    _kind = _ffi.ACQ_EI
    _n_theta_samples = 3


class maPI(_ClosedForm):
    _kind = _ffi.ACQ_PI
    _n_theta_samples = 10

    def __init__(self, *a, **kw):
        super(maPI, self).__init__(*a, **kw)
        self.jitter = 1e-6


class EI(maEI):
    """EI.py: single-output specialisation (n_hyps_samples = 1, :35)."""

    def __init__(self, *a, **kw):
        super(EI, self).__init__(*a, **kw)
        self.n_hyps_samples = 1


class PI(maPI):
    """PI.py: single-output specialisation.  Unlike EI.py:35 it keeps n_hyps_samples = min(10, number_of_hyps_samples())
    (PI.py:34): the h-loop averages over the hyper-samples and leaves the model on the last one."""


class _MonteCarlo(AcquisitionBase):
    """uEI_noiseless.py:9-175 / uPI.py:9-125: Monte-Carlo EI / PI of a composite utility
    U(theta, f(x)) with common random numbers W."""
    _kind = _ffi.ACQ_EI

    def __init__(self, model, space, optimizer=None, cost_withGradients=None, utility=None):
        self.optimizer = optimizer
        self.utility = utility
        super(_MonteCarlo, self).__init__(model, space, optimizer, cost_withGradients=cost_withGradients)
        if cost_withGradients is not None:
            print('LBC acquisition does now make sense with cost. Cost set to constant.')
        self.cost_withGradients = constant_cost_withGradients
        self.n_attributes = self.model.output_dim
        self.W_samples = np.random.normal(size=(25, self.n_attributes))          # uEI_noiseless.py:31
        self.n_hyps_samples = min(10, self.model.number_of_hyps_samples())
        self.use_full_support = self.utility.parameter_dist.use_full_support
        if self.use_full_support:
            self.utility_params_samples = self.utility.parameter_dist.support
            self.utility_prob_dist = np.atleast_1d(self.utility.parameter_dist.prob_dist)
        else:
            self.utility_params_samples = self.utility.parameter_dist.sample(10)  # uEI_noiseless.py:38

    def _thetas(self):
        s = np.asarray(self.utility_params_samples, dtype=float)
        return s.reshape(len(self.utility_params_samples), -1)

    def _compute_acq(self, X, parallel=True):
        """`parallel` is accepted for signature compatibility: the reference's pathos variant
        (uEI_noiseless.py:85-116) computes the same numbers one candidate at a time."""
        X = np.atleast_2d(X)
        model = self._device_model()
        kind = self._device_kind_or_none()
        if kind is None:
            return self._host_utility_acq(X)
        prob = self.utility_prob_dist if self.use_full_support else None
        thetas = self._thetas()
        if kind in (_ffi.UTIL_NEG_SUM_EXP, _ffi.UTIL_NEG_EXP_COS):
            thetas = np.zeros((thetas.shape[0], 1))        # parameter unused by these utilities
        acqX = model.acq_mc(X, self._kind, kind, self.utility.device_params, thetas, prob, W=self.W_samples, n_hyps=self.n_hyps_samples)
        return np.reshape(acqX, (X.shape[0], 1))

    def _compute_acq_withGradients(self, X):
        """uEI_noiseless.py:118-136: full support -> the support and its weights; otherwise ONE freshly
        sampled theta per call (parameter_dist.sample(1), a global-RNG touchpoint)."""
        if not type(self).analytical_gradient_prediction:
            raise NotImplementedError('')
        X = np.atleast_2d(X)
        if self.use_full_support:
            samples2, prob = self.utility.parameter_dist.support, self.utility_prob_dist
        else:
            samples2, prob = self.utility.parameter_dist.sample(1), None
        kind = self._device_kind_or_none()
        if kind is None:
            return self._host_utility_acq_with_gradients(X, samples2, prob)
        thetas = np.asarray(samples2, dtype=float).reshape(len(samples2), -1)
        if kind in (_ffi.UTIL_NEG_SUM_EXP, _ffi.UTIL_NEG_EXP_COS):
            thetas = np.zeros((thetas.shape[0], 1))
        acqX, dacq_dX = self._device_model().acq_mc_grad(X, kind, self.utility.device_params, thetas, prob, W=self.W_samples,
                                                         n_hyps=self.n_hyps_samples)
        return np.reshape(acqX, (X.shape[0], 1)), np.reshape(dacq_dX, X.shape)

    # ---- utilities outside the device's closed set (utility.py:37-41 accepts ANY callable): the posterior still comes from
    # the device (K*, the N^2 C contraction, gradients -- all the O(N^2) work), only U itself is evaluated on the host with the
    # user's func / dfunc ("closed enum + host fallback for anything else"); nothing here is test infrastructure.
    def _device_kind_or_none(self):
        try:
            return self.utility.device_kind(self.model.output_dim)
        except NotImplementedError:
            if not getattr(self, "_warned_host_utility", False):
                warnings.warn("bocf_amd: the utility is a Python callable outside the device's closed set (%s): the posterior "
                              "(mean, variance, gradients) is computed on the GPU, the Monte-Carlo loop over U runs on the HOST -- "
                              "orders of magnitude slower than a device utility (Utility(..., device=...))"
                              % ", ".join(sorted(["linear", "neg_sq_dist", "neg_sum_exp", "neg_exp_cos", "rosenbrock"])), RuntimeWarning,
                              stacklevel=3)
                self._warned_host_utility = True
            return None

    def _eval_user_utility(self, f, theta, y):
        """f(theta, y) for y (m, n): one vectorised call when the callable broadcasts over candidates the way the reference
        itself uses it on the (m, N) evaluated points (uEI_noiseless.py:76), else candidate by candidate.  Returns (n,) -- or
        (m, n) for a gradient."""
        n = y.shape[1]
        if getattr(self, "_user_vectorised", None) is not False:
            try:
                out = np.asarray(f(theta, y), dtype=float)
                if out.shape in ((n,), (1, n), (y.shape[0], n)):
                    if getattr(self, "_user_vectorised", None) is None:        # check the broadcast once against single columns
                        k = min(n, 3)
                        one = np.stack([np.asarray(f(theta, y[:, i]), dtype=float).reshape(-1) for i in range(k)], -1)
                        ok = np.allclose(one.reshape(-1, k), out.reshape(-1, n)[:, :k], rtol=1e-12, atol=1e-300)
                        self._user_vectorised = bool(ok)
                    if self._user_vectorised:
                        return out.reshape(-1, n) if out.ndim == 2 and out.shape[0] == y.shape[0] and out.shape[0] > 1 else out.reshape(n)
            except Exception:
                self._user_vectorised = False
        cols = [np.asarray(f(theta, y[:, i]), dtype=float).reshape(-1) for i in range(n)]
        out = np.stack(cols, -1)
        return out[0] if out.shape[0] == 1 else out

    def _host_utility_acq(self, X):
        """uEI_noiseless.py:63-83 / uPI.py:66-86 with the user's callable: best-so-far from the hyper-sample current on entry
        (:66), then per hyper-sample the device posterior and the host Monte-Carlo sum."""
        model = self._device_model()
        thetas = list(self.utility_params_samples)
        prob = self.utility_prob_dist if self.use_full_support else None
        n, L = X.shape[0], len(thetas)
        marg = np.zeros((n, L))
        f_eval = model.posterior_mean_at_evaluated_points()
        best = [float(np.max(self._eval_user_utility(self.utility.eval_func, th, f_eval))) for th in thetas]
        pi = self._kind == _ffi.ACQ_PI
        W = np.asarray(self.W_samples, dtype=float)
        for h in range(self.n_hyps_samples):
            model.set_hyperparameters(h)
            mu = model.posterior_mean(X)
            sigma = np.sqrt(model.posterior_variance(X))
            for l, th in enumerate(thetas):
                for w in W:
                    val = self._eval_user_utility(self.utility.eval_func, th, mu + sigma * w[:, None])
                    if pi:
                        marg[:, l] += (val - (best[l] + 1e-6)) > 0.0          # uPI.py:83
                    else:
                        marg[:, l] += np.maximum(val - best[l], 0.0)          # uEI_noiseless.py:80
        marg /= (self.n_hyps_samples * W.shape[0])
        acq = marg.dot(np.atleast_1d(prob)) if prob is not None else marg.sum(1) / L
        return acq.reshape(n, 1)

    def _host_utility_acq_with_gradients(self, X, thetas, prob):
        """uEI_noiseless.py:138-170 with the user's func and dfunc (device posterior + gradients, host Monte-Carlo loop)."""
        if self.utility.dfunc is None:
            raise TypeError("the utility has no dfunc: the gradient of the acquisition cannot be computed (utility.py:44-48)")
        model = self._device_model()
        thetas = list(thetas)
        n, d, L = X.shape[0], X.shape[1], len(thetas)
        marg, dmarg = np.zeros((n, L)), np.zeros((n, d, L))
        f_eval = model.posterior_mean_at_evaluated_points()
        best = [float(np.max(self._eval_user_utility(self.utility.eval_func, th, f_eval))) for th in thetas]
        W = np.asarray(self.W_samples, dtype=float)
        for h in range(self.n_hyps_samples):
            model.set_hyperparameters(h)
            mu = model.posterior_mean(X)
            sigma = np.sqrt(model.posterior_variance(X))
            dmu = model.posterior_mean_gradient(X)                            # (m, n, d)
            dvar = model.posterior_variance_gradient(X)
            for l, th in enumerate(thetas):
                for w in W:
                    a = mu + sigma * w[:, None]
                    val = self._eval_user_utility(self.utility.eval_func, th, a)
                    marg[:, l] += np.maximum(val - best[l], 0.0)
                    imp = val > best[l]
                    if imp.any():
                        g = np.asarray(self._eval_user_utility(self.utility.eval_gradient, th, a[:, imp]), dtype=float).reshape(mu.shape[0], -1)
                        b = dmu[:, imp, :] + (0.5 * w[:, None] / sigma[:, imp])[:, :, None] * dvar[:, imp, :]
                        dmarg[imp, :, l] += np.einsum("ji,jiq->iq", g, b)
        marg /= (self.n_hyps_samples * W.shape[0])
        dmarg /= (self.n_hyps_samples * W.shape[0])
        if prob is not None:
            acq, dacq = marg.dot(np.atleast_1d(prob)), np.tensordot(dmarg, np.atleast_1d(prob), 1)
        else:
            acq, dacq = marg.sum(1) / L, dmarg.sum(2) / L
        return acq.reshape(n, 1), dacq.reshape(n, d)

    def update_Z_samples(self, n_samples):
        """uEI_noiseless.py:172-175.  `n_samples` is REQUIRED as in the reference: cbo.py:299-302 calls
        update_Z_samples() without it inside a try/except, so in a cbo run the TypeError is swallowed, W_samples is
        never redrawn and np.random is not consumed -- a seeded cbo-style trajectory relies on exactly that."""
        print('Update utility parameter W and Z samples')
        self.W_samples = np.random.normal(size=self.W_samples.shape)


class uEI_noiseless(_MonteCarlo):
    analytical_gradient_prediction = True
    _kind = _ffi.ACQ_EI


class uPI(_MonteCarlo):
    analytical_gradient_prediction = False
    _kind = _ffi.ACQ_PI

    def __init__(self, *a, **kw):
        super(uPI, self).__init__(*a, **kw)
        self.jitter = 1e-6
