"""Acquisition plug-ins with the reference's class surface (GPyOpt/acquisitions/base.py:5-74,
maEI.py, maPI.py, EI.py, PI.py, uEI_noiseless.py, uPI.py) whose `_compute_acq` runs on the
MI355X: predict (K* build, N^2 C contraction) + acquisition in one device pass per call.

The host keeps exactly the reference's RNG touchpoints (W_samples drawn with np.random.normal
at construction, theta sampled with np.random.choice) so seeded trajectories line up.
There is no CPU fallback: the model must be a bocf_amd.multi_outputGP.
"""
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
        kind = self.utility.device_kind(self.model.output_dim)
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
        kind = self.utility.device_kind(self.model.output_dim)
        thetas = np.asarray(samples2, dtype=float).reshape(len(samples2), -1)
        if kind in (_ffi.UTIL_NEG_SUM_EXP, _ffi.UTIL_NEG_EXP_COS):
            thetas = np.zeros((thetas.shape[0], 1))
        acqX, dacq_dX = self._device_model().acq_mc_grad(X, kind, self.utility.device_params, thetas, prob, W=self.W_samples,
                                                         n_hyps=self.n_hyps_samples)
        return np.reshape(acqX, (X.shape[0], 1)), np.reshape(dacq_dX, X.shape)

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
