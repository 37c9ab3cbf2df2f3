"""MI355X-backed drop-in for the reference's `multi_outputGP` (multi_outputGP.py:9-349), both with fixed
hyper-parameters (GPModelFixedHyps, GPyOpt/models/gpmodel_fixed_hyps.py:9-120) and with learned ones (GPModel,
GPyOpt/models/gpmodel.py:9-230: optimise + HMC per update, n_samples hyper-samples resident on the device).

Same constructor, attributes and method set; every number is produced by libbocf_hip.so
(no CPU fallback).  The m independent GPs are fitted and evaluated together on the device:
results are stacked as (m, n) float64 arrays exactly like the reference.
"""
import ctypes

import numpy as np

from . import _ffi
from .hyper import OutputHyper
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
    :param kernel: list of kernels (bocf_amd.kern.* or duck-typed GPy kernels).  None entries: fixed_hyps ->
        SE(variance=2, lengthscale=0.3) (gpmodel_fixed_hyps.py:50); otherwise SE(variance=1, ARD=ARD[j]) (gpmodel.py:58).
    :param noise_var: list of noise variances.  None entries: fixed_hyps -> 1e-10 (gpmodel_fixed_hyps.py:56); otherwise
        0.01 Var(Y) as the starting value of a free noise (gpmodel.py:64).
    :param exact_feval: list of bools; with learning, True fixes the noise at 1e-6 (gpmodel.py:71-72).
    :param ARD: list of bools (default all True, multi_outputGP.py:44-47); used for default kernels when learning.
    :param n_samples: number of hyper-parameter samples (number_of_hyps_samples()).
    :param fixed_hyps: True -> GPModelFixedHyps semantics (one set of hyper-parameters, set_hyperparameters a no-op).
        False (the reference's default) -> GPModel semantics: every updateModel optimises the hyper-parameters, runs
        HMC and keeps n_samples draws (gpmodel.py:102-128); all inferences run on the device, the m outputs in
        lockstep (bocf_amd/hyper.py).  The sampler settings are the attributes n_burnin, subsample_interval,
        step_size, leapfrog_steps, max_iters (GPModel's defaults, gpmodel.py:32).
    :param device: HIP device index (default: LOCAL_RANK or 0).
    """
    analytical_gradient_prediction = True

    def __init__(self, output_dim, kernel=None, noise_var=None, exact_feval=None, n_samples=10, ARD=None, fixed_hyps=False,
                 device=None, reference_instance_kernels=False):
        self.output_dim = output_dim
        # fixed_hyps=False with a USER kernel: the reference learns the hyper-parameters with that kernel but builds its n_samples
        # prediction instances as SE whatever it was (gpmodel.py:57-61 resets self.kernel to None, :80-84 then takes the SE branch) and
        # writes the HMC samples into them (:121-126).  False (default): the instances keep the family that was learned.  True: the
        # reference's behaviour, bit for bit in the family it predicts with.
        self.reference_instance_kernels = bool(reference_instance_kernels)
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
        # ---- hyper-parameter learning (fixed_hyps=False): GPModel's sampler settings (gpmodel.py:32)
        self.n_burnin, self.subsample_interval, self.step_size, self.leapfrog_steps, self.max_iters = 100, 10, 1e-1, 20, 200
        # a leapfrog trajectory whose Ky stops factorizing: "raise" = the reference (LinAlgError out of jitchol propagates out of
        # updateModel, GPy/util/linalg.py:71 <- gpmodel.py:117-118); "reject" = drop that proposal for that output and carry on
        self.hmc_on_failure = "raise"
        self.device_hmc_streamed = True         # N > 128 (or d > 16): the stream-resident chain (bocf_hmc_streamed); False = lockstep host loop
        self.device_hmc = True                  # N <= 128, d <= 16: the whole HMC chain in one device launch (bocf_hmc); False = lockstep host loop
        self._H = 1 if fixed_hyps else int(n_samples)     # hyper-samples resident on the device
        self._current_h = 0                                # set_hyperparameters(h)
        self._sampler_outputs = None                       # per output: parameter state of GPModel.model
        self._Ymat = None                                  # (m, N) targets, cached for the inferences of one update
        self._ibuf = None                                  # argument block of bocf_infer (arrays + ctypes pointers)
        self._query_cache = self._grad_cache = None        # last all-hyper-sample posterior query (served per h as slices)
        self._fit_serial = 0
        self._instances = None                             # [h][j] -> (variance, lengthscale (d,), noise): GPModel.model_instances
        self._kernel_ids = None
        self.hmc_samples = None
        self.last_update_info = {}
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
        st["_query_cache"] = st["_grad_cache"] = None
        st["_ibuf"] = None
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
        if name in ("shard_fit", "shard_fit_simulate") and value:
            # an output-sharded fit exchanges the inverse factors only: the O(N^2) incremental updates need the upper factor,
            # which stays on its owner, so every updateModel is a (sharded) refit
            self.incremental = False

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
        self._Ymat = None
        self._ibuf = None
        if not self.fixed_hyps:
            return self._update_hyper_samples()
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
        kids, var, ls, noise = self._hyper_arrays()
        return (tuple(kids), var.tobytes(), ls.tobytes(), noise.tobytes())

    def _hyper_arrays(self):
        """(kernel ids (M), variance (M,), lengthscale (M, d), noise (M,)) of the M = H * m factorizations resident on the
        device, hyper-sample-major (H = 1 with fixed hyper-parameters).  The outputs may use different kernel families (the
        reference takes a kernel list, multi_outputGP.py:44-47): the ids go to the device with bocf_set_kernel_ids."""
        d = self._X.shape[1]
        if not self.fixed_hyps:
            if self._instances is None:
                raise RuntimeError("updateModel has not been called")
            flat = [inst for group in self._instances for inst in group]
            inst_ids = [_ffi.KERN_SE] * self.output_dim if self.reference_instance_kernels else list(self._kernel_ids)      # gpmodel.py:80-84
            return (inst_ids * len(self._instances), _ffi.f64([v for v, _, _ in flat]), _ffi.f64([l for _, l, _ in flat]),
                    _ffi.f64([n for _, _, n in flat]))
        kids, var, ls = [], [], []
        for j in range(self.output_dim):
            k = self.kernel[j] if self.kernel[j] is not None else SE(d, variance=2., lengthscale=0.3)
            kj, vj, lj = kernel_spec(k, d)
            kids.append(int(kj))
            var.append(vj)
            ls.append(lj)
        noise = [1e-10 if nv is None else float(nv) for nv in self.noise_var]
        return kids, _ffi.f64(var), _ffi.f64(ls), _ffi.f64(noise)

    def _send_kernel_ids(self, kids):
        """Outputs of different kernel families: hand the id list to the device for the NEXT fit / inference / chain (it is consumed by
        that call); returns the scalar kernel id argument of that call."""
        kids = [int(k) for k in kids]
        if len(set(kids)) > 1:
            arr = (ctypes.c_int * len(kids))(*kids)
            _ffi.check(_ffi.load().bocf_set_kernel_ids(self._context().handle, arr, len(kids)), "bocf_set_kernel_ids")
        return kids[0]

    def _device_fit(self, kids, var, ls, noise, groups):
        """One bocf_fit over var.size factorizations (`groups` copies of the m targets); returns (jitter, lml)."""
        lib, ctx = _ffi.load(), self._context()
        kid = self._send_kernel_ids(kids)
        N, d = self._X.shape
        M = var.size
        Y = _ffi.f64(np.tile(np.stack([y[:, 0] for y in self._Y], 0), (groups, 1)))
        jit, lml = np.zeros(M), np.zeros(M)
        rc = lib.bocf_fit(ctx.handle, _ffi.dptr(self._X), _ffi.dptr(Y), N, d, M, kid, _ffi.dptr(var), _ffi.dptr(ls), _ffi.dptr(noise), 5,
                          _ffi.dptr(jit), _ffi.dptr(lml))
        _ffi.check(rc, "bocf_fit")
        self._fitted = False
        self._W_key = None
        self._cand_token = None
        if rc > 0:   # jitchol gave up (GPy/util/linalg.py:71) for the outputs whose last rung still has a bad pivot
            err = np.linalg.LinAlgError("not positive definite, even with jitter.")
            err.outputs = self._failed_outputs(M)
            raise err
        return jit, lml

    def _failed_outputs(self, M):
        """Outputs whose factorization failed in the last bocf_fit / bocf_infer (the library's own per-output info)."""
        import ctypes
        info = (ctypes.c_int * M)()
        _ffi.check(_ffi.load().bocf_last_fit_info(self._context().handle, info, M), "bocf_last_fit_info")
        return [j for j in range(M) if info[j] != 0]

    def _fit(self):
        kids, var, ls, noise = self._hyper_arrays()
        self._context().set_option("hyper_samples", self._H)
        self.jitter, self.log_marginal = self._device_fit(kids, var, ls, noise, self._H)
        self._fit_key = self._hyper_key()
        self._fitted = True
        self._fit_serial += 1
        self._query_cache = self._grad_cache = None

    # ---- hyper-parameter learning: GPModel._create_model / updateModel (gpmodel.py:50-128) ------------------------
    def _create_sampler_state(self):
        d = self._X.shape[1]
        self._sampler_outputs, self._instance_noise, kids = [], [], []
        for j in range(self.output_dim):
            k = self.kernel[j]
            if k is None:
                k = SE(d, variance=1., ARD=bool(self.ARD[j]))                        # gpmodel.py:58
            # (with a user kernel the reference builds its model_instances from SE all the same, gpmodel.py:80-84 -- a
            #  slip: self.kernel was reset to None at :61; here the instances keep the user's kernel family unless
            #  reference_instance_kernels=True asks for the reference's behaviour)
            kj, vj, lj = kernel_spec(k, d)
            kids.append(int(kj))
            ard = bool(getattr(k, "ARD", np.asarray(k.lengthscale).size > 1))
            if self.exact_feval[j]:
                noise, fixed = 1e-6, True                                            # gpmodel.py:71-72
            elif self.noise_var[j] is not None:
                noise, fixed = float(self.noise_var[j]), True                        # :73-74
            else:
                noise, fixed = float(np.var(self._Y[j])) * 0.01, False               # :64, :75-76
            self._sampler_outputs.append(OutputHyper(vj, lj if ard else lj[:1], noise, fixed))
            self._instance_noise.append(noise)
        self._kernel_ids = kids

    def _infer_buffers(self):
        """Argument block of bocf_infer, built once per data set: thousands of inferences reuse the same arrays and
        ctypes pointers."""
        N, d = self._X.shape
        m = self.output_dim
        key = (id(self._X), N, d, m)
        b = self._ibuf
        if b is None or b["key"] != key:
            if self._Ymat is None:
                self._Ymat = _ffi.f64(np.stack([y[:, 0] for y in self._Y], 0))
            hyp = np.zeros((m, d + 2))                     # [variance, lengthscale (d), noise] per output, one validity check
            arr = dict(var=np.zeros(m), ls=np.zeros((m, d)), noise=np.zeros(m), jit=np.zeros(m), lml=np.zeros(m), dv=np.zeros(m),
                       dl=np.zeros((m, d)), dn=np.zeros(m))
            b = dict(key=key, hyp=hyp, X=self._X, Y=self._Ymat, **arr)
            b["args"] = (_ffi.dptr(self._X), _ffi.dptr(self._Ymat), N, d, m, self._kernel_ids[0], _ffi.dptr(arr["var"]), _ffi.dptr(arr["ls"]),
                         _ffi.dptr(arr["noise"]), 5, _ffi.dptr(arr["jit"]), _ffi.dptr(arr["lml"]), _ffi.dptr(arr["dv"]), _ffi.dptr(arr["dl"]),
                         _ffi.dptr(arr["dn"]))
            self._ibuf = b
        return b

    def _infer(self, params):
        """One batched device inference for the m sampler models: log-marginals and their hyper-gradients."""
        b = self._infer_buffers()
        var, ls, noise, hyp = b["var"], b["ls"], b["noise"], b["hyp"]
        for j, (v, l, nz) in enumerate(params):
            var[j] = v
            ls[j] = l
            noise[j] = nz
        hyp[:, 0] = var
        hyp[:, 1:-1] = ls
        hyp[:, -1] = noise + 1.0                           # noise may be 0: shift it into the "> 0" test
        with np.errstate(invalid="ignore"):
            if not (np.isfinite(hyp).all() and (hyp[:, :-1] > 0).all() and (noise >= 0).all()):
                ok = np.isfinite(hyp).all(axis=1) & (hyp[:, :-1] > 0).all(axis=1) & (noise >= 0)
                err = np.linalg.LinAlgError("hyper-parameters left the positive domain")
                err.outputs = [int(j) for j in np.flatnonzero(~ok)]
                raise err
        self._send_kernel_ids(self._kernel_ids)             # (no-op unless the outputs differ in kernel family)
        rc = _ffi.load().bocf_infer(self._context().handle, *b["args"])
        _ffi.check(rc, "bocf_infer")
        self._fitted = False
        self._W_key = None
        self._cand_token = None
        if rc > 0:   # jitchol gave up (GPy/util/linalg.py:71) for the outputs whose last rung still has a bad pivot
            err = np.linalg.LinAlgError("not positive definite, even with jitter.")
            err.outputs = self._failed_outputs(self.output_dim)
            raise err
        return b["lml"].copy(), b["dv"].copy(), b["dl"].copy(), b["dn"].copy()

    def _update_hyper_samples(self):
        from .hyper import LockstepSampler
        if self._sampler_outputs is None:
            self._create_sampler_state()
        outs, d = self._sampler_outputs, self._X.shape[1]
        sampler = LockstepSampler(outs, self._infer, d, device_hmc=self._device_hmc if self.device_hmc else None)
        ctx = self._context()
        sampler.evaluate()                       # first inference uploads X, Y; the thousands that follow reuse them
        ctx.set_option("reuse_data", 1)
        ctx.set_option("skip_mu_train", 1)
        try:
            opt_info, n_opt, num_samples, chains = self._optimize_and_sample(sampler, outs)
        finally:
            ctx.set_option("reuse_data", 0)
            ctx.set_option("skip_mu_train", 0)
        self.hmc_samples = [ch[self.n_burnin::self.subsample_interval] for ch in chains]                              # :119
        self._instances = []
        for h in range(self.n_samples):                                               # :121-126
            group = []
            for j, o in enumerate(outs):
                theta = o.param_array.copy()
                theta[-1] = self._instance_noise[j]        # a fixed noise keeps its constrained value in the instances
                theta[~o.fixed] = self.hmc_samples[j][h]
                ls = theta[1:-1]
                group.append((theta[0], np.full(d, ls[0]) if ls.size == 1 else ls.copy(), theta[-1]))
            self._instances.append(group)
        self.last_update_info = dict(optimizer_inferences=n_opt, hmc_inferences=sampler.n_inferences - n_opt,
                                     optimizer_iterations=opt_info["iterations"], accepted=sampler.accepted.copy(), num_samples=num_samples)
        self._fit()
        self._current_h = 0                                                           # :128

    def _device_hmc(self, outs, momenta, uniforms, hmc_iters, stepsize, raise_on_failure):
        """The chain of every output on the device (hmc.py:30-69): ONE launch for models with N <= 128, d <= 16 (bocf_hmc), the
        stream-resident chain beyond that (bocf_hmc_streamed: one inference's launches per leapfrog step, no host round trip).
        Returns (chains, accepted, diverged, n_inferences, status, draws_done); draws_done < len(uniforms[0]) means the streamed chain
        met a factorization that needs jitchol's ladder in that draw -- the caller runs it on the host and calls again for the rest.
        None when the model is outside what the device chains serve (outputs with different parameter counts)."""
        N, d = self._X.shape
        m = len(outs)
        sizes = {o.param_array.size for o in outs}
        if len(sizes) != 1:
            return None
        P = sizes.pop()
        nls = P - 2
        if nls not in (1, d):
            return None
        fused = N <= 128 and d <= 16
        if not fused and not self.device_hmc_streamed:
            return None
        ns = len(uniforms[0])
        theta = _ffi.f64(np.stack([o.param_array for o in outs]))
        fixed = np.ascontiguousarray(np.stack([o.fixed for o in outs]).astype(np.int32))
        mom = np.zeros((m, ns, P))
        for j in range(m):
            pf = int(np.sum(~outs[j].fixed))
            mom[j, :, :pf] = np.asarray(momenta[j], dtype=float).reshape(ns, pf)
        uni = _ffi.f64(np.stack([np.asarray(u, dtype=float) for u in uniforms]))
        if self._Ymat is None:
            self._Ymat = _ffi.f64(np.stack([y[:, 0] for y in self._Y], 0))
        chains = np.zeros((m, ns, P))
        acc, div, status = np.zeros(m, dtype=np.int32), np.zeros(m, dtype=np.int32), np.zeros(m, dtype=np.int32)
        ninf = ctypes.c_longlong(0)
        done = ctypes.c_int(ns)
        ip = ctypes.POINTER(ctypes.c_int)
        pr = outs[0].prior
        lib, ctx = _ffi.load(), self._context()
        kid = self._send_kernel_ids(self._kernel_ids)
        if fused:
            rc = lib.bocf_hmc(ctx.handle, _ffi.dptr(self._X), _ffi.dptr(self._Ymat), N, d, m, kid, _ffi.dptr(theta), nls,
                              fixed.ctypes.data_as(ip), pr.a, pr.b, _ffi.dptr(mom), _ffi.dptr(uni), ns, int(hmc_iters), float(stepsize), 5,
                              1 if raise_on_failure else 0, _ffi.dptr(chains), acc.ctypes.data_as(ip), div.ctypes.data_as(ip),
                              status.ctypes.data_as(ip), ctypes.byref(ninf))
            _ffi.check(rc, "bocf_hmc")
        else:
            rc = lib.bocf_hmc_streamed(ctx.handle, _ffi.dptr(self._X), _ffi.dptr(self._Ymat), N, d, m, kid, _ffi.dptr(theta), nls,
                                       fixed.ctypes.data_as(ip), pr.a, pr.b, _ffi.dptr(mom), _ffi.dptr(uni), ns, int(hmc_iters), float(stepsize),
                                       _ffi.dptr(chains), acc.ctypes.data_as(ip), div.ctypes.data_as(ip), ctypes.byref(done), ctypes.byref(ninf))
            _ffi.check(rc, "bocf_hmc_streamed")
        self._fitted = False
        self._W_key = None
        self._cand_token = None
        for j, o in enumerate(outs):
            o.param_array[:] = theta[j]
        out_chains = [chains[j, :, :int(np.sum(~outs[j].fixed))].copy() for j in range(m)]
        return out_chains, acc.astype(int), div.astype(int), ninf.value, status, int(done.value)

    def _optimize_and_sample(self, sampler, outs):
        from .hyper import LockstepSampler
        opt_info = sampler.optimize(self.max_iters)                                   # gpmodel.py:115
        n_opt = sampler.n_inferences
        num_samples = self.n_burnin + self.n_samples * self.subsample_interval
        draws = LockstepSampler.draw(outs, num_samples)
        for o, (eps, _, _) in zip(outs, draws):
            o.param_array[:] = o.param_array * (1. + eps * 0.01)                      # :116 (raw write: a fixed noise moves too)
        chains = sampler.hmc([dr[1] for dr in draws], [dr[2] for dr in draws], self.leapfrog_steps, self.step_size,
                             on_failure=self.hmc_on_failure)                          # :117-118
        return opt_info, n_opt, num_samples, chains

    def number_of_hyps_samples(self):
        return self.n_samples

    def set_hyperparameters(self, n):
        """gpmodel_fixed_hyps.py:76-77: no-op with fixed hyper-parameters; gpmodel.py:137-138: select hyper-sample n --
        every posterior query below answers for that sample (all n_samples factorizations stay on the device)."""
        if not self.fixed_hyps:
            if not 0 <= int(n) < self._H:
                raise IndexError("hyper-sample %r out of range (n_samples = %d)" % (n, self._H))
            self._current_h = int(n)

    def _rows(self):
        """Rows of the device's (H * m, ...) results that belong to the current hyper-sample."""
        return slice(self._current_h * self.output_dim, (self._current_h + 1) * self.output_dim)

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
        key = None
        if self._H > 1:
            # cbo.py walks the hyper-samples with set_hyperparameters(h) + a posterior query per h (cbo.py:162-166,176-178):
            # the device answers for all H at once, so the queries after the first are slices of the same pass
            self._ensure_fitted()
            key = ("p", flags, want_var, np.shape(X), hash(np.ascontiguousarray(X, dtype=np.float64).tobytes()), self._fit_serial)
            if self._query_cache is not None and self._query_cache[0] == key:
                mean, var = self._query_cache[1]
                return mean[self._rows()].copy(), (var[self._rows()].copy() if want_var else None)
        n = self._set_candidates(X)
        M = self.output_dim * self._H
        mean = np.empty((M, n))
        var = np.empty((M, n)) if want_var else None
        if n:
            _ffi.check(_ffi.load().bocf_predict(self._context().handle, flags, _ffi.dptr(mean), _ffi.dptr(var)), "bocf_predict")
        if self._H > 1:
            self._query_cache = (key, (mean, var))
            return mean[self._rows()].copy(), (var[self._rows()].copy() if want_var else None)
        return mean, var

    def predict(self, X, full_cov=False):
        """Posterior means and variances at X, likelihood noise included, clipped at 1e-10
        (multi_outputGP.py:138-149 -> gpmodel_fixed_hyps.py:79-87).  Returns ((m, n), (m, n))."""
        if full_cov:
            # each output's model returns its n x n covariance and the wrapper keeps column 0 of it (multi_outputGP.py:146-148):
            # row j = the covariances of every point with the FIRST one, noise on entry 0, every entry clipped at 1e-10
            X = np.atleast_2d(X)
            mean, _ = self._predict(X, 0, want_var=False)
            M = self.output_dim * self._H
            cov = np.empty((M, X.shape[0]))
            if self._set_candidates(X):         # (the mean may have come from the hyper-sample cache without touching the device)
                _ffi.check(_ffi.load().bocf_predict_cov_column(self._context().handle, _ffi.ADD_NOISE | _ffi.CLIP, _ffi.dptr(cov)),
                           "bocf_predict_cov_column")
            return mean, cov[self._rows()].copy()
        return self._predict(np.atleast_2d(X), _ffi.ADD_NOISE | _ffi.CLIP)

    def predict_noiseless(self, X, full_cov=False):
        """multi_outputGP.py:151-162 -> gpmodel_fixed_hyps.py:89-97 / gpmodel.py:151-159 (full_cov is accepted and never looked at)."""
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
        out = np.empty((self.output_dim * self._H, self._X.shape[0]))
        _ffi.check(_ffi.load().bocf_mean_at_train(self._context().handle, _ffi.dptr(out)), "bocf_mean_at_train")
        return out[self._rows()].copy() if self._H > 1 else out

    def _gradients(self, X):
        key = None
        if self._H > 1:
            self._ensure_fitted()
            key = ("g", np.shape(X), hash(np.ascontiguousarray(np.atleast_2d(X), dtype=np.float64).tobytes()), self._fit_serial)
            if self._grad_cache is not None and self._grad_cache[0] == key:
                dmean, dvar = self._grad_cache[1]
                return dmean[self._rows()].copy(), dvar[self._rows()].copy()
        n = self._set_candidates(np.atleast_2d(X))
        d = self._X.shape[1]
        M = self.output_dim * self._H
        dmean, dvar = np.empty((M, n, d)), np.empty((M, n, d))
        if n:
            _ffi.check(_ffi.load().bocf_predict_gradients(self._context().handle, _ffi.dptr(dmean), _ffi.dptr(dvar)), "bocf_predict_gradients")
        if self._H > 1:
            self._grad_cache = (key, (dmean, dvar))
            return dmean[self._rows()].copy(), dvar[self._rows()].copy()
        return dmean, dvar

    def posterior_mean_gradient(self, X):
        """d mu / dX, (m, n, d)  (multi_outputGP.py:284-294 -> gp.py:438-461)."""
        return self._gradients(X)[0]

    def posterior_variance_gradient(self, X):
        """d var / dX, (m, n, d)  (multi_outputGP.py:297-306 -> gp.py:464-490)."""
        return self._gradients(X)[1]

    # ---- reference methods with no caller on the path (multi_outputGP.py:204-281,309-330: knowledge-gradient style
    # look-ahead helpers that cbo.py and the EI/PI acquisitions never reach): named so that a script using them fails
    # with a clear message instead of an AttributeError
    def _off_path(self, *a, **kw):
        raise NotImplementedError("not part of the accelerated path: cbo.py and the maEI/maPI/uEI_noiseless/uPI/EI/PI acquisitions never "
                                  "call it (SURVEY.md section 8b)")

    partial_precomputation_for_covariance = partial_precomputation_for_covariance_gradient = _off_path
    partial_precomputation_for_variance_conditioned_on_next_point = posterior_variance_conditioned_on_next_point = _off_path
    posterior_variance_gradient_conditioned_on_next_point = posterior_covariance_between_points = _off_path
    posterior_covariance_between_points_partially_precomputed = posterior_covariance_gradient = _off_path
    posterior_covariance_gradient_partially_precomputed = _off_path

    def set_hyperparameters2(self, hyperparameters):
        """multi_outputGP.py:118-120: per-output hyper-sample indices; the device keeps hyper-samples aligned across
        outputs (sample h of every output forms model h), so only a common index is supported."""
        idx = {int(h) for h in hyperparameters}
        if len(idx) != 1:
            raise NotImplementedError("per-output hyper-sample indices must be equal on the device path")
        self.set_hyperparameters(idx.pop())

    def get_hyperparameters_samples(self, n_samples=1):
        """multi_outputGP.py:123-128 -> gpmodel.py: the retained HMC draws, [sample][output] -> unfixed parameters."""
        if self.fixed_hyps or self.hmc_samples is None:
            return [[None] * self.output_dim for _ in range(n_samples)]
        return [[self.hmc_samples[j][i].copy() for j in range(self.output_dim)] for i in range(min(n_samples, self._H))]

    # ---- inspection ----------------------------------------------------------------------------
    def get_factor(self, j):
        """(L lower (N, N), alpha (N,)) of output j -- Posterior.woodbury_chol / woodbury_vector."""
        self._ensure_fitted()
        N = self._X.shape[0]
        L, a = np.empty((N, N)), np.empty(N)
        _ffi.check(_ffi.load().bocf_get_factor(self._context().handle, self._current_h * self.output_dim + j, _ffi.dptr(L), _ffi.dptr(a)),
                   "bocf_get_factor")
        return L, a

    def log_likelihood(self):
        """Log marginal likelihood per output (GP.log_likelihood, gp.py:262-266), (m,)."""
        self._ensure_fitted()
        return self.log_marginal[self._rows()].copy()

    def log_likelihood_gradients(self):
        """d log-marginal / d (kernel variance (m,), lengthscales (m, d), noise variance (m,)) of the current fit --
        what GP.parameters_changed leaves in kern.variance.gradient / kern.lengthscale.gradient /
        likelihood.variance.gradient (gp.py:256-258), in raw (untransformed) parameters.  For an isotropic
        kernel sum the lengthscale gradients over d."""
        self._ensure_fitted()
        m, d = self.output_dim * self._H, self._X.shape[1]
        dv, dl, dn = np.empty(m), np.empty((m, d)), np.empty(m)
        _ffi.check(_ffi.load().bocf_lml_gradients(self._context().handle, _ffi.dptr(dv), _ffi.dptr(dl), _ffi.dptr(dn)), "bocf_lml_gradients")
        return dv[self._rows()], dl[self._rows()], dn[self._rows()]

    def get_train_kernel(self, j):
        self._ensure_fitted()
        N = self._X.shape[0]
        K = np.empty((N, N))
        _ffi.check(_ffi.load().bocf_get_train_kernel(self._context().handle, self._current_h * self.output_dim + j, _ffi.dptr(K)),
                   "bocf_get_train_kernel")
        return K

    def get_model_parameters(self):
        """multi_outputGP.py:333-339: per output [variance, lengthscale..., noise]."""
        _, var, ls, noise = self._hyper_arrays()
        r = self._rows()
        var, ls, noise = var[r], ls[r], noise[r]
        return [np.atleast_2d(np.concatenate(([var[j]], ls[j], [noise[j]]))) for j in range(self.output_dim)]

    def get_model_parameters_names(self):
        d = self._X.shape[1]
        names = ["variance"] + ["lengthscale_%d" % q for q in range(d)] + ["Gaussian_noise.variance"]
        return [list(names) for _ in range(self.output_dim)]

    # ---- fused acquisition entry points used by bocf_amd.acquisitions ---------------------------
    def _begin_acq(self, n_hyps, own_best):
        """The reference's h-loop (maEI.py:85-97, uEI_noiseless.py:71-82) runs inside the device call: tell it how many
        hyper-samples to average (n_hyps_samples = min(10, number_of_hyps_samples())) and whose best-so-far to use --
        each hyper-sample's own (maEI.py:88) or the one that is current on entry (uEI_noiseless.py:66) -- and leave the
        model on the loop's last hyper-sample as set_hyperparameters(h) does."""
        self._ensure_fitted()
        if self.fixed_hyps:
            return
        n_h = self._H if n_hyps is None else max(1, min(int(n_hyps), self._H))
        ctx = self._context()
        ctx.set_option("acq_hyper_samples", n_h)
        ctx.set_option("best_group", -1 if own_best else self._current_h)
        self._current_h = n_h - 1

    def acq_linear(self, X, kind, thetas, prob, n_hyps=None):
        """Closed-form EI/PI of theta.f over the batch X on the device (bocf_acq_linear)."""
        self._begin_acq(n_hyps, True)
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

    def acq_linear_grad(self, X, kind, thetas, prob, n_hyps=None):
        """(acq (n,), d acq/dX (n, d)) of the closed-form EI/PI (bocf_acq_linear_grad)."""
        self._begin_acq(n_hyps, True)
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

    def acq_mc_grad(self, X, util_kind, util_params, thetas, prob, W=None, n_hyps=None):
        """(acq (n,), d acq/dX (n, d)) of the Monte-Carlo EI (bocf_acq_mc_grad)."""
        self._begin_acq(n_hyps, False)
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

    def acq_mc(self, X, kind, util_kind, util_params, thetas, prob, W=None, fetch=True, n_hyps=None):
        """Monte-Carlo EI/PI of a device utility over the batch X (bocf_acq_mc)."""
        self._begin_acq(n_hyps, False)
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
