"""MI355X-backed drop-in for the reference's `multi_outputGP` (multi_outputGP.py:9-349) with
fixed hyper-parameters (GPModelFixedHyps, GPyOpt/models/gpmodel_fixed_hyps.py:9-120).

Same constructor, attributes and method set; every number is produced by libbocf_hip.so
(no CPU fallback).  The m independent GPs are fitted and evaluated together on the device:
results are stacked as (m, n) float64 arrays exactly like the reference.
"""
import numpy as np

from . import _ffi
from .kern import SE, kernel_spec


class _GPView(object):
    """What `output[j].model` exposes in the reference (GPRegression): the data."""

    def __init__(self, parent, j):
        self._p, self._j = parent, j

    @property
    def X(self):
        return self._p._X

    @property
    def Y(self):
        return self._p._Y[self._j]


class _OutputView(object):
    """Per-output facade with the GPModelFixedHyps method set (gpmodel_fixed_hyps.py:61-120);
    each call evaluates all outputs on the device and returns row j as an (n, 1) column."""
    analytical_gradient_prediction = True

    def __init__(self, parent, j):
        self._p, self._j = parent, j
        self.model = _GPView(parent, j)

    def _col(self, a):
        return a[self._j][:, None].copy()

    def predict(self, X, full_cov=False):
        m, v = self._p.predict(X, full_cov)
        return self._col(m), self._col(v)

    def predict_noiseless(self, X, full_cov=False):
        m, v = self._p.predict_noiseless(X, full_cov)
        return self._col(m), self._col(v)

    def posterior_mean(self, X):
        return self._col(self._p.posterior_mean(np.atleast_2d(X)))

    def posterior_variance(self, X):
        return self._col(self._p.posterior_variance(np.atleast_2d(X)))

    def posterior_variance_noiseless(self, X):
        return self._col(self._p.posterior_variance_noiseless(np.atleast_2d(X)))

    def posterior_mean_gradient(self, X):     # gpmodel_fixed_hyps.py:187-192 -> (n, d)
        return self._p.posterior_mean_gradient(np.atleast_2d(X))[self._j].copy()

    def posterior_variance_gradient(self, X):
        return self._p.posterior_variance_gradient(np.atleast_2d(X))[self._j].copy()

    def set_hyperparameters(self, i):
        pass

    def get_fmin(self):                       # gpmodel_fixed_hyps.py:181-185
        return self._p.predict(self._p._X)[0][self._j].min()


class multi_outputGP(object):
    """
    General class for handling a multi-output Gaussian process (drop-in for multi_outputGP.py:9).

    :param output_dim: number of outputs.
    :param kernel: list of kernels (bocf_amd.kern.* or duck-typed GPy kernels); None entries use
        the GPModelFixedHyps default SE(variance=2, lengthscale=0.3) (gpmodel_fixed_hyps.py:50).
    :param noise_var: list of noise variances; None entries -> 1e-10 (gpmodel_fixed_hyps.py:56).
    :param exact_feval, ARD: accepted for signature compatibility (unused by GPModelFixedHyps).
    :param n_samples: value returned by number_of_hyps_samples().
    :param fixed_hyps: must be True -- hyper-parameter learning (GPModel.updateModel's
        optimize + HMC, gpmodel.py:102-128) is a later row of the scope table.
    :param device: HIP device index (default: LOCAL_RANK or 0).
    """
    analytical_gradient_prediction = True

    def __init__(self, output_dim, kernel=None, noise_var=None, exact_feval=None, n_samples=10, ARD=None, fixed_hyps=False,
                 device=None):
        if not fixed_hyps:
            raise NotImplementedError("bocf_amd.multi_outputGP covers the fixed-hyper-parameter path (fixed_hyps=True); "
                                      "hyper-parameter learning (optimize + HMC) is not built on the device yet")
        self.output_dim = output_dim
        self.kernel = [None] * output_dim if kernel is None else list(kernel)
        self.noise_var = [None] * output_dim if noise_var is None else list(noise_var)
        self.exact_feval = [False] * output_dim if exact_feval is None else exact_feval
        self.n_samples = n_samples
        self.ARD = [True] * output_dim if ARD is None else ARD
        self.fixed_hyps = fixed_hyps
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self._ctx = None
        self._X = None
        self._Y = None
        self._fitted = False
        self._cand_token = None
        self._fit_key = None
        self._W_key = None
        self.incremental = True       # O(N^2) updateModel when only targets change or one observation is appended
        self.jitter = None
        self.log_marginal = None
        self.output = [_OutputView(self, j) for j in range(output_dim)]

    # ---- device handle management (objects are pickled into pathos workers in the reference,
    # acquisition_optimizer.py:131-133: the handle is dropped and rebuilt lazily)
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_ctx"] = None
        st["_fitted"] = False
        st["_cand_token"] = None
        st["_W_key"] = None
        return st

    def _context(self):
        if self._ctx is None:
            self._ctx = _ffi.Context(self.device)
        return self._ctx

    def _ensure_fitted(self):
        if self._X is None:
            raise RuntimeError("updateModel has not been called")
        if not self._fitted:
            self._fit()

    def set_option(self, name, value):
        self._context().set_option(name, value)

    # ---- fit ---------------------------------------------------------------------------------
    def updateModel(self, X_all, Y_all):
        """Updates the model with new observations (multi_outputGP.py:97-102): X_all (N, d),
        Y_all list of m arrays (N, 1)."""
        X = _ffi.f64(np.atleast_2d(X_all))
        Y = [np.asarray(y, dtype=np.float64).reshape(-1) for y in Y_all]
        if len(Y) != self.output_dim or any(y.shape[0] != X.shape[0] for y in Y):
            raise ValueError("Y_all must hold output_dim arrays of N observations")
        prevX = self._X
        self._X, self._Y = X.copy(), [y[:, None].copy() for y in Y]
        if self.incremental and self._fitted and prevX is not None and self._hyper_key() == self._fit_key:
            # cbo.py adds one observation per iteration (cbo.py:363,419) and the reference refits from scratch
            # (GP.set_XY, gp.py:191-227); with the factor resident on the device the two common cases are O(N^2)
            Ymat = _ffi.f64(np.stack(Y, 0))
            lib, ctx = _ffi.load(), self._context()
            lml = np.zeros(self.output_dim)
            if X.shape == prevX.shape and np.array_equal(X, prevX):
                _ffi.check(lib.bocf_update_targets(ctx.handle, _ffi.dptr(Ymat), _ffi.dptr(lml)), "bocf_update_targets")
                self.log_marginal = lml
                return
            if X.shape[0] == prevX.shape[0] + 1 and X.shape[1] == prevX.shape[1] and np.array_equal(X[:-1], prevX):
                xnew = _ffi.f64(X[-1])
                rc = _ffi.check(lib.bocf_append(ctx.handle, _ffi.dptr(xnew), _ffi.dptr(Ymat), _ffi.dptr(lml)), "bocf_append")
                if rc == 0:
                    self.log_marginal = lml
                    return
        self._fit()

    def _hyper_key(self):
        kid, var, ls, noise = self._hyper_arrays()
        return (kid, var.tobytes(), ls.tobytes(), noise.tobytes())

    def _hyper_arrays(self):
        d = self._X.shape[1]
        kid, var, ls = None, [], []
        for j in range(self.output_dim):
            k = self.kernel[j] if self.kernel[j] is not None else SE(d, variance=2., lengthscale=0.3)
            kj, vj, lj = kernel_spec(k, d)
            if kid is None:
                kid = kj
            elif kid != kj and {kid, kj} != {_ffi.KERN_RBF, _ffi.KERN_SE}:
                raise NotImplementedError("all outputs must use the same kernel family on the device")
            var.append(vj)
            ls.append(lj)
        noise = [1e-10 if nv is None else float(nv) for nv in self.noise_var]
        return kid, _ffi.f64(var), _ffi.f64(ls), _ffi.f64(noise)

    def _fit(self):
        lib, ctx = _ffi.load(), self._context()
        kid, var, ls, noise = self._hyper_arrays()
        N, d = self._X.shape
        m = self.output_dim
        Y = _ffi.f64(np.stack([y[:, 0] for y in self._Y], 0))
        jit, lml = np.zeros(m), np.zeros(m)
        rc = lib.bocf_fit(ctx.handle, _ffi.dptr(self._X), _ffi.dptr(Y), N, d, m, kid, _ffi.dptr(var), _ffi.dptr(ls), _ffi.dptr(noise), 5,
                          _ffi.dptr(jit), _ffi.dptr(lml))
        _ffi.check(rc, "bocf_fit")
        if rc > 0:   # jitchol gave up (GPy/util/linalg.py:71)
            raise np.linalg.LinAlgError("not positive definite, even with jitter.")
        self.jitter, self.log_marginal = jit, lml
        self._W_key = None
        self._fit_key = self._hyper_key()
        self._fitted = True
        self._cand_token = None

    def number_of_hyps_samples(self):
        return self.n_samples

    def set_hyperparameters(self, n):          # gpmodel_fixed_hyps.py:76-77
        pass

    def get_evaluated_points(self):
        return np.copy(self._X)

    # ---- predictions ---------------------------------------------------------------------------
    def _set_candidates(self, X):
        self._ensure_fitted()
        X = _ffi.f64(X)
        if X.ndim != 2 or X.shape[1] != self._X.shape[1]:
            raise ValueError("candidates must be (n, %d)" % self._X.shape[1])
        _ffi.check(_ffi.load().bocf_set_candidates(self._context().handle, _ffi.dptr(X), X.shape[0]), "bocf_set_candidates")
        return X.shape[0]

    def _predict(self, X, flags, want_var=True):
        n = self._set_candidates(X)
        mean = np.empty((self.output_dim, n))
        var = np.empty((self.output_dim, n)) if want_var else None
        if n:
            _ffi.check(_ffi.load().bocf_predict(self._context().handle, flags, _ffi.dptr(mean), _ffi.dptr(var)), "bocf_predict")
        return mean, var

    def predict(self, X, full_cov=False):
        """Posterior means and variances at X, likelihood noise included, clipped at 1e-10
        (multi_outputGP.py:138-149 -> gpmodel_fixed_hyps.py:79-87).  Returns ((m, n), (m, n))."""
        if full_cov:
            raise NotImplementedError("full_cov is not part of the accelerated path")
        return self._predict(np.atleast_2d(X), _ffi.ADD_NOISE | _ffi.CLIP)

    def predict_noiseless(self, X, full_cov=False):
        """multi_outputGP.py:151-162 -> gpmodel_fixed_hyps.py:89-97."""
        if full_cov:
            raise NotImplementedError("full_cov is not part of the accelerated path")
        return self._predict(np.atleast_2d(X), _ffi.CLIP)

    def posterior_mean(self, X):
        """multi_outputGP.py:165-173."""
        return self._predict(X, 0, want_var=False)[0]

    def posterior_variance(self, X):
        """Noise included, clipped (multi_outputGP.py:183-191 -> gpmodel_fixed_hyps.py:106-112)."""
        return self._predict(X, _ffi.ADD_NOISE | _ffi.CLIP)[1]

    def posterior_variance_noiseless(self, X):
        """multi_outputGP.py:194-200."""
        return self._predict(X, _ffi.CLIP)[1]

    def posterior_mean_at_evaluated_points(self):
        """multi_outputGP.py:176-180; cached on the device at fit time."""
        self._ensure_fitted()
        out = np.empty((self.output_dim, self._X.shape[0]))
        _ffi.check(_ffi.load().bocf_mean_at_train(self._context().handle, _ffi.dptr(out)), "bocf_mean_at_train")
        return out

    def _gradients(self, X):
        n = self._set_candidates(np.atleast_2d(X))
        d = self._X.shape[1]
        dmean, dvar = np.empty((self.output_dim, n, d)), np.empty((self.output_dim, n, d))
        if n:
            _ffi.check(_ffi.load().bocf_predict_gradients(self._context().handle, _ffi.dptr(dmean), _ffi.dptr(dvar)), "bocf_predict_gradients")
        return dmean, dvar

    def posterior_mean_gradient(self, X):
        """d mu / dX, (m, n, d)  (multi_outputGP.py:284-294 -> gp.py:438-461)."""
        return self._gradients(X)[0]

    def posterior_variance_gradient(self, X):
        """d var / dX, (m, n, d)  (multi_outputGP.py:297-306 -> gp.py:464-490)."""
        return self._gradients(X)[1]

    # ---- inspection ----------------------------------------------------------------------------
    def get_factor(self, j):
        """(L lower (N, N), alpha (N,)) of output j -- Posterior.woodbury_chol / woodbury_vector."""
        self._ensure_fitted()
        N = self._X.shape[0]
        L, a = np.empty((N, N)), np.empty(N)
        _ffi.check(_ffi.load().bocf_get_factor(self._context().handle, j, _ffi.dptr(L), _ffi.dptr(a)), "bocf_get_factor")
        return L, a

    def log_likelihood(self):
        """Log marginal likelihood per output (GP.log_likelihood, gp.py:262-266), (m,)."""
        self._ensure_fitted()
        return self.log_marginal.copy()

    def log_likelihood_gradients(self):
        """d log-marginal / d (kernel variance (m,), lengthscales (m, d), noise variance (m,)) of the current fit --
        what GP.parameters_changed leaves in kern.variance.gradient / kern.lengthscale.gradient /
        likelihood.variance.gradient (gp.py:256-258), in raw (untransformed) parameters.  For an isotropic
        kernel sum the lengthscale gradients over d."""
        self._ensure_fitted()
        m, d = self.output_dim, self._X.shape[1]
        dv, dl, dn = np.empty(m), np.empty((m, d)), np.empty(m)
        _ffi.check(_ffi.load().bocf_lml_gradients(self._context().handle, _ffi.dptr(dv), _ffi.dptr(dl), _ffi.dptr(dn)), "bocf_lml_gradients")
        return dv, dl, dn

    def get_train_kernel(self, j):
        self._ensure_fitted()
        N = self._X.shape[0]
        K = np.empty((N, N))
        _ffi.check(_ffi.load().bocf_get_train_kernel(self._context().handle, j, _ffi.dptr(K)), "bocf_get_train_kernel")
        return K

    def get_model_parameters(self):
        """multi_outputGP.py:333-339: per output [variance, lengthscale..., noise]."""
        _, var, ls, noise = self._hyper_arrays()
        return [np.atleast_2d(np.concatenate(([var[j]], ls[j], [noise[j]]))) for j in range(self.output_dim)]

    def get_model_parameters_names(self):
        d = self._X.shape[1]
        names = ["variance"] + ["lengthscale_%d" % q for q in range(d)] + ["Gaussian_noise.variance"]
        return [list(names) for _ in range(self.output_dim)]

    # ---- fused acquisition entry points used by bocf_amd.acquisitions ---------------------------
    def acq_linear(self, X, kind, thetas, prob):
        """Closed-form EI/PI of theta.f over the batch X on the device (bocf_acq_linear)."""
        n = self._set_candidates(np.atleast_2d(X))
        thetas = _ffi.f64(np.atleast_2d(thetas))
        if thetas.shape[1] != self.output_dim:
            raise ValueError("theta must have output_dim entries")
        prob = None if prob is None else _ffi.f64(np.atleast_1d(prob))
        acq = np.empty(n)
        if n:
            _ffi.check(_ffi.load().bocf_acq_linear(self._context().handle, kind, _ffi.dptr(thetas), _ffi.dptr(prob), thetas.shape[0],
                                                   _ffi.dptr(acq)), "bocf_acq_linear")
        return acq

    def acq_linear_grad(self, X, kind, thetas, prob):
        """(acq (n,), d acq/dX (n, d)) of the closed-form EI/PI (bocf_acq_linear_grad)."""
        n = self._set_candidates(np.atleast_2d(X))
        thetas = _ffi.f64(np.atleast_2d(thetas))
        if thetas.shape[1] != self.output_dim:
            raise ValueError("theta must have output_dim entries")
        prob = None if prob is None else _ffi.f64(np.atleast_1d(prob))
        acq, dacq = np.empty(n), np.empty((n, self._X.shape[1]))
        if n:
            _ffi.check(_ffi.load().bocf_acq_linear_grad(self._context().handle, kind, _ffi.dptr(thetas), _ffi.dptr(prob), thetas.shape[0],
                                                        _ffi.dptr(acq), _ffi.dptr(dacq)), "bocf_acq_linear_grad")
        return acq, dacq

    def acq_mc_grad(self, X, util_kind, util_params, thetas, prob, W=None):
        """(acq (n,), d acq/dX (n, d)) of the Monte-Carlo EI (bocf_acq_mc_grad)."""
        if W is not None:
            self.set_mc_samples(W)
        n = self._set_candidates(np.atleast_2d(X))
        params = None if util_params is None else _ffi.f64(np.atleast_1d(util_params))
        th = _ffi.f64(np.atleast_2d(thetas))
        L, tdim = th.shape
        prob = None if prob is None else _ffi.f64(np.atleast_1d(prob))
        acq, dacq = np.empty(n), np.empty((n, self._X.shape[1]))
        if n:
            _ffi.check(_ffi.load().bocf_acq_mc_grad(self._context().handle, util_kind, _ffi.dptr(params), 0 if params is None else params.size,
                                                    _ffi.dptr(th), tdim, _ffi.dptr(prob), L, _ffi.dptr(acq), _ffi.dptr(dacq)), "bocf_acq_mc_grad")
        return acq, dacq

    def set_mc_samples(self, W):
        self._ensure_fitted()
        W = _ffi.f64(np.atleast_2d(W))
        if W.shape[1] != self.output_dim:
            raise ValueError("W must be (S, output_dim)")
        key = (W.shape, hash(W.tobytes()))
        if key == self._W_key:       # the same common random numbers are already resident (L-BFGS calls f_df hundreds of times)
            return
        _ffi.check(_ffi.load().bocf_set_mc_samples(self._context().handle, _ffi.dptr(W), W.shape[0]), "bocf_set_mc_samples")
        self._W_key = key

    def acq_mc(self, X, kind, util_kind, util_params, thetas, prob, W=None, fetch=True):
        """Monte-Carlo EI/PI of a device utility over the batch X (bocf_acq_mc)."""
        if W is not None:
            self.set_mc_samples(W)
        n = self._set_candidates(np.atleast_2d(X)) if X is not None else None
        return self._acq_mc_resident(kind, util_kind, util_params, thetas, prob, n, fetch)

    def _acq_mc_resident(self, kind, util_kind, util_params, thetas, prob, n, fetch=True):
        lib, ctx = _ffi.load(), self._context()
        params = None if util_params is None else _ffi.f64(np.atleast_1d(util_params))
        if thetas is None:
            th, L, tdim = None, 1, 0
        else:
            th = _ffi.f64(np.atleast_2d(thetas))
            L, tdim = th.shape
        prob = None if prob is None else _ffi.f64(np.atleast_1d(prob))
        acq = np.empty(n) if (fetch and n is not None) else None
        if n is None or n > 0:
            _ffi.check(lib.bocf_acq_mc(ctx.handle, kind, util_kind, _ffi.dptr(params), 0 if params is None else params.size, _ffi.dptr(th),
                                       tdim, _ffi.dptr(prob), L, _ffi.dptr(acq)), "bocf_acq_mc")
        return acq

    def select_topk(self, k):
        """(indices, values) of the k best candidates of the last acquisition call -- the
        np.argsort(-acq)[:k] of anchor_points_generator.py:61, ties to the lowest index."""
        idx = np.empty(k, dtype=np.int64)
        val = np.empty(k)
        _ffi.check(_ffi.load().bocf_select_topk(self._context().handle, k, idx.ctypes.data_as(_ffi._c_ll_p), _ffi.dptr(val)),
                   "bocf_select_topk")
        keep = idx >= 0
        return idx[keep], val[keep]
