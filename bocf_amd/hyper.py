"""Hyper-parameter learning of the reference's GPModel (GPyOpt/models/gpmodel.py:50-128) with every GP inference on the
MI355X.

Per model update and per output the reference runs, on the CPU and one output after the other,
    model.optimize(max_iters=200)                      <= 200 inferences  (paramz L-BFGS-B on the transformed parameters)
    param_array *= 1 + 0.01 randn                      (gpmodel.py:116)
    HMC(stepsize=.1).sample(200 draws, 20 leapfrog)    4000 inferences    (GPy/inference/mcmc/hmc.py:30-69)
    keep draws [100::10] -> 10 hyper-samples           (gpmodel.py:118-119)
where one inference = kernel matrix + Cholesky + alpha + log-marginal + dL/dK + kernel hyper-gradients.  Here the m
outputs advance in LOCKSTEP: their chains are independent, so one device call (`bocf_fit` + `bocf_lml_gradients`, all
outputs batched, each with its own current hyper-parameters) serves one leapfrog step -- or one quasi-Newton trial
point -- of every output.  What stays on the host is O(m * P) arithmetic: priors, the Logexp transform, momenta,
the Metropolis test.  The host RNG is consumed in exactly the reference's order (all of output 0's draws, then output
1's ...): the number of draws per output is fixed, so they are drawn up front.

paramz (==0.9.1, requirements.txt:15) is not part of the reference tree; the pieces of it on this path are restated
from its published source: transformations.Logexp, Model._transform_gradients / optimizer_array / _objective_grads
and optimization.opt_lbfgsb's stopping parameters.
"""
import numpy as np

from .acquisition_optimizer import lbfgsb_batched

_LIM = 36.0
_LOG_LIM = np.log(np.finfo(np.float64).max)


# ---- paramz.transformations.Logexp ---------------------------------------------------------------------------------
def logexp_f(x):
    x = np.asarray(x, dtype=float)
    return np.where(x > _LIM, x, np.log1p(np.exp(np.clip(x, -_LOG_LIM, _LIM))))


def logexp_finv(f):
    f = np.asarray(f, dtype=float)
    with np.errstate(over="ignore"):
        return np.where(f > _LIM, f, np.log(np.expm1(f)))


def logexp_gradfactor(f, df):
    return df * np.where(f > _LIM, 1.0, -np.expm1(-f))


def logexp_log_jacobian(f):
    return logexp_finv(f) - f


def logexp_log_jacobian_grad(f):
    return 1.0 / np.expm1(f)


def _lgamma(a):
    import math
    return math.lgamma(a)


class Gamma(object):
    """GPy/core/parameterization/priors.py:264-330 (a shape, b rate)."""

    def __init__(self, a, b):
        self.a, self.b = float(a), float(b)
        self.constant = -_lgamma(self.a) + a * np.log(b)

    @staticmethod
    def from_EV(E, V):
        return Gamma(np.square(E) / V, E / V)

    def lnpdf(self, x):
        return self.constant + (self.a - 1) * np.log(x) - self.b * x

    def lnpdf_grad(self, x):
        return (self.a - 1.) / x - self.b


class OutputHyper(object):
    """Parameter state of the `self.model` of one GPModel (gpmodel.py:50-75): [kern.variance, kern.lengthscale (1 or d
    entries), Gaussian_noise.variance], Gamma.from_EV(2, 4) prior on each (:67-68), Logexp constraint on each except a
    noise that is constrain_fixed (exact_feval -> 1e-6, or a given noise_var; :71-76)."""

    def __init__(self, variance, lengthscale, noise_var, noise_fixed, prior=None):
        ls = np.atleast_1d(np.asarray(lengthscale, dtype=float))
        self.param_array = np.concatenate(([float(variance)], ls, [float(noise_var)]))
        self.fixed = np.zeros(self.param_array.size, dtype=bool)
        self.fixed[-1] = bool(noise_fixed)
        self.prior = prior if prior is not None else Gamma.from_EV(2., 4.)

    @property
    def unfixed_param_array(self):
        return self.param_array[~self.fixed].copy()

    @property
    def optimizer_array(self):
        return logexp_finv(self.param_array[~self.fixed])

    @optimizer_array.setter
    def optimizer_array(self, x):
        self.param_array[~self.fixed] = logexp_f(x)

    def expanded(self, d):
        """(variance, lengthscale (d,), noise) for the device (an isotropic lengthscale is repeated)."""
        p = self.param_array
        ls = p[1:-1]
        return float(p[0]), (np.full(d, ls[0]) if ls.size == 1 else ls.copy()), float(p[-1])

    def log_prior(self):                                   # priorizable.py:49-65
        x = self.param_array
        return float(np.sum(self.prior.lnpdf(x)) + np.sum(logexp_log_jacobian(x[~self.fixed])))

    def log_prior_gradients(self):                         # priorizable.py:67-82
        x = self.param_array
        ret = self.prior.lnpdf_grad(x)
        ret[~self.fixed] += logexp_log_jacobian_grad(x[~self.fixed])
        return ret

    def transform_gradients(self, g):                      # paramz Model._transform_gradients
        free = ~self.fixed
        return logexp_gradfactor(self.param_array[free], np.asarray(g, dtype=float)[free])


class LockstepSampler(object):
    """Optimise + HMC for all outputs together.  `infer(params)` with params = [(variance, ls (d,), noise)] * m returns
    (lml (m,), dvar (m,), dls (m, d), dnoise (m,)) of ONE batched device inference and raises np.linalg.LinAlgError
    when a factorization fails even with jitter (GPy/util/linalg.py:71)."""

    def __init__(self, outputs, infer, d, device_hmc=None):
        """device_hmc (optional): callable(outputs, momenta, uniforms, hmc_iters, stepsize, raise_on_failure) that runs the WHOLE chain of
        every output on the device (bocf_hmc: one launch; models with N <= 128, d <= 16) and returns (chains, accepted, diverged,
        n_inferences, status) -- or None when the model is outside what the device chain serves, in which case hmc() advances the
        chains on the host, one batched device inference per leapfrog step."""
        self.outputs, self.infer, self.d = list(outputs), infer, int(d)
        self.device_hmc = device_hmc
        self.n_inferences = 0
        self._key = None
        self._obj = self._tgrad = None
        self._safe = None                  # per output: last parameters whose inference succeeded
        sizes = {o.param_array.size for o in self.outputs}
        priors = {(o.prior.a, o.prior.b) for o in self.outputs}
        self._uniform = len(sizes) == 1 and len(priors) == 1
        self._free = np.stack([~o.fixed for o in self.outputs]) if self._uniform else None
        self.failed = np.zeros(len(self.outputs), dtype=bool)
        self.factor_failed = np.zeros(len(self.outputs), dtype=bool)   # subset of `failed`: jitchol gave up (not a mere overflow)

    # -- objective = -(log-marginal + log-prior) and its gradient w.r.t. optimizer_array, all outputs (model.py:72-104)
    def evaluate(self):
        """(objective (m,), [gradient w.r.t. optimizer_array] * m) at the outputs' current parameters.  An output whose
        factorization fails (LinAlgError carrying `.outputs`) gets +inf / zero gradient and `self.failed[j]`; the batch is
        re-run with that output's last good parameters in its slot so the other outputs still get their values.  (The
        reference, one output at a time, turns such a failure into +inf inside the optimiser -- paramz
        Model._objective_grads -- and lets it propagate out of HMC.)"""
        key = b"".join(o.param_array.tobytes() for o in self.outputs)
        if key != self._key:
            m = len(self.outputs)
            params = [o.expanded(self.d) for o in self.outputs]
            failed = np.zeros(m, dtype=bool)
            while True:
                try:
                    lml, dvar, dls, dnoise = self.infer(params)
                    self.n_inferences += 1
                    break
                except np.linalg.LinAlgError as e:
                    newly = [j for j in getattr(e, "outputs", range(m)) if not failed[j]]
                    if not newly or self._safe is None:
                        raise
                    for j in newly:
                        failed[j] = True
                        params[j] = self._safe[j]
            factor_failed = failed.copy()
            obj, tg = self._objective_terms(lml, dvar, dls, dnoise, failed)
            if self._safe is None:
                self._safe = list(params)
            for j in range(m):
                if not failed[j]:
                    self._safe[j] = params[j]
            self._key, self._obj, self._tgrad, self.failed, self.factor_failed = key, obj, tg, failed, factor_failed
        return self._obj, self._tgrad

    def _objective_terms(self, lml, dvar, dls, dnoise, failed):
        """-(log-marginal + log-prior) and its gradient w.r.t. optimizer_array per output; outputs that share the
        parameter count and the prior (the usual case) are handled in one set of array operations (same elementwise
        functions and summation order as the per-output path, so both give the same bits)."""
        outs, m = self.outputs, len(self.outputs)
        obj, tg = np.empty(m), [None] * m
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            if self._uniform:
                TH = np.stack([o.param_array for o in outs])
                FREE = self._free
                pr = outs[0].prior
                nls = TH.shape[1] - 2
                DL = dls if nls == self.d else np.sum(dls, axis=1, keepdims=True)
                G = np.concatenate((np.asarray(dvar, dtype=float)[:, None], DL, np.asarray(dnoise, dtype=float)[:, None]), axis=1)
                expm1 = np.expm1(TH)
                big = TH > _LIM
                with np.errstate(divide="ignore"):
                    logj = np.where(FREE, np.where(big, TH, np.log(expm1)) - TH, 0.0)
                lp = np.sum(pr.lnpdf(TH), axis=1) + np.sum(logj, axis=1)
                pg = pr.lnpdf_grad(TH) + np.where(FREE, 1.0 / expm1, 0.0)
                TG = -(G + pg) * np.where(big, 1.0, -np.expm1(-TH))
                for j in range(m):
                    obj[j] = -float(lml[j]) - float(lp[j])
                    tg[j] = TG[j][FREE[j]]
            else:
                for j, o in enumerate(outs):
                    nls = o.param_array.size - 2
                    dl = dls[j] if nls == self.d else np.array([np.sum(dls[j])])       # isotropic: one shared lengthscale
                    g = np.concatenate(([dvar[j]], dl, [dnoise[j]]))
                    obj[j] = -float(lml[j]) - o.log_prior()
                    tg[j] = o.transform_gradients(-(g + o.log_prior_gradients()))
        for j in range(m):
            if failed[j] or not (np.isfinite(obj[j]) and np.all(np.isfinite(tg[j]))):   # overflowed inference: same as a failed one
                failed[j] = True
                obj[j] = np.inf
                tg[j] = np.zeros(int(np.sum(~outs[j].fixed)))
        return obj, tg

    def _restore(self, j, obj_j, tgrad_j):
        """Output j went back to a state whose objective / gradient are known (rejected proposal): patch the cache
        instead of running another inference."""
        obj, tg = self._obj.copy(), list(self._tgrad)
        obj[j], tg[j] = obj_j, tgrad_j
        self._key = b"".join(o.param_array.tobytes() for o in self.outputs)
        self._obj, self._tgrad = obj, tg

    # -- model.optimize(max_iters=200): gpmodel.py:115 -> paramz opt_lbfgsb (factr 1e7, pgtol 1e-5, maxfun = maxiter)
    def optimize(self, max_iters=200):
        m = len(self.outputs)
        P = max(o.optimizer_array.size for o in self.outputs)
        X0 = np.zeros((m, P))
        for j, o in enumerate(self.outputs):
            X0[j, :o.optimizer_array.size] = o.optimizer_array
        def f_df(Z, rows):
            for z, j in zip(Z, rows):
                o = self.outputs[j]
                with np.errstate(over="ignore", invalid="ignore"):
                    o.optimizer_array = z[:o.optimizer_array.size]
            obj, tg = self.evaluate()
            F, G = np.full(len(rows), np.inf), np.zeros((len(rows), P))
            for i, j in enumerate(rows):
                if not self.failed[j]:            # a failed inference is +inf to the optimiser (paramz Model._objective_grads)
                    F[i] = obj[j]
                    G[i, :tg[j].size] = tg[j]
            return F, G
        info = {}
        X, _ = lbfgsb_batched(f_df, X0, [(-np.inf, np.inf)] * P, maxiter=max_iters, maxfun=max_iters, factr=1e7, pgtol=1e-5,
                              with_rows=True, info=info)
        for j, o in enumerate(self.outputs):
            o.optimizer_array = X[j, :o.optimizer_array.size]
        return info

    # -- GPy/inference/mcmc/hmc.py:30-69 with M = I, all outputs in lockstep
    @staticmethod
    def draw(outputs, num_samples, rng=np.random):
        """The host-RNG draws of gpmodel.py:116-118 in the reference's order: for each output in turn, randn(P_all) for
        the 1 % jitter, then per HMC sample one multivariate_normal(0, I_{P_free}) and one rand()."""
        draws = []
        for o in outputs:
            P = int(np.sum(~o.fixed))
            eps = rng.randn(o.param_array.size)
            mom, u = np.empty((num_samples, P)), np.empty(num_samples)
            for i in range(num_samples):
                mom[i] = rng.multivariate_normal(np.zeros(P), np.eye(P))
                u[i] = rng.rand()
            draws.append((eps, mom, u))
        return draws

    def hmc(self, momenta, uniforms, hmc_iters=20, stepsize=1e-1, on_failure="raise"):
        """momenta[j] (num_samples, P_j), uniforms[j] (num_samples,).  Returns the per-output chains
        (num_samples, P_j) of unfixed parameters, recorded as hmc.py:45-59 does (current state, overwritten on accept).

        A trajectory can leave the region where Ky factorizes.  The reference does not catch that: jitchol's LinAlgError
        (GPy/util/linalg.py:71) propagates out of HMC.sample and GPModel.updateModel (gpmodel.py:117-118) -- on_failure =
        "raise" (default) does the same.  "reject" treats such a proposal as rejected for that output and carries on (the
        other outputs' chains are unaffected).  A merely non-finite objective (overflow) is NOT an exception in the
        reference either: H_new = nan fails the Metropolis test (hmc.py:51-58) -- rejected in both modes."""
        if on_failure not in ("raise", "reject"):
            raise ValueError("on_failure must be 'raise' or 'reject'")
        outs = self.outputs
        m, num_samples = len(outs), len(uniforms[0])
        chains = [np.empty((num_samples, int(np.sum(~o.fixed)))) for o in outs]
        self.accepted = np.zeros(m, dtype=int)
        self.diverged = np.zeros(m, dtype=int)
        i = 0
        while i < num_samples:
            if self.device_hmc is not None and self._uniform:
                res = self.device_hmc(outs, [mo[i:] for mo in momenta], [u[i:] for u in uniforms], hmc_iters, stepsize, on_failure == "raise")
                if res is None:
                    self.device_hmc = None                        # outside what the device chains serve: host loop from here on
                    continue
                part, acc, div, n_inf, status, done = res
                self.n_inferences += int(n_inf)
                self._key = None                                  # the outputs moved: the cached objective is stale
                if np.any(status != 0):                           # jitchol gave up inside a trajectory (hmc.py lets it propagate)
                    err = np.linalg.LinAlgError("not positive definite, even with jitter.")
                    err.outputs = [int(j) for j in np.flatnonzero(status != 0)]
                    raise err
                for j in range(m):
                    chains[j][i:i + done] = part[j][:done]
                self.accepted += acc
                self.diverged += div
                i += done
                if i >= num_samples:
                    break
                # draw i needs jitchol's ladder (or left the domain): this ONE draw on the host, then the device takes over again
            self._host_draw(i, chains, momenta, uniforms, hmc_iters, stepsize, on_failure)
            i += 1
        return chains

    def _host_draw(self, i, chains, momenta, uniforms, hmc_iters, stepsize, on_failure):
        """Draw i of every output with one batched device inference per leapfrog step (hmc.py:43-66)."""
        outs = self.outputs
        m = len(outs)
        half_log_2pi = np.log(2 * np.pi) / 2.
        with np.errstate(over="ignore", invalid="ignore"):
            obj, tg = self.evaluate()
            p = [momenta[j][i].copy() for j in range(m)]
            H_old = [obj[j] + p[j].size * half_log_2pi + np.dot(p[j], p[j]) / 2. for j in range(m)]   # log det I = 0
            x_old = [o.optimizer_array for o in outs]
            x = [v.copy() for v in x_old]
            old = [(obj[j], tg[j]) for j in range(m)]
            for j, o in enumerate(outs):
                chains[j][i] = o.param_array[~o.fixed]
            diverged = np.zeros(m, dtype=bool)                # trajectory left the domain where Ky factorizes: rejected
            for _ in range(hmc_iters):                        # hmc.py:62-66
                for j, o in enumerate(outs):
                    p[j] += (-stepsize / 2.) * tg[j]
                    x[j] += stepsize * p[j]                   # (the reference re-reads optimizer_array = finv(f(x)) here)
                    o.optimizer_array = x[j]
                obj, tg = self.evaluate()
                if on_failure == "raise" and self.factor_failed.any():
                    err = np.linalg.LinAlgError("not positive definite, even with jitter.")
                    err.outputs = [int(j) for j in np.flatnonzero(self.factor_failed)]
                    for j in err.outputs:                     # leave the model where the draw started
                        outs[j].optimizer_array = x_old[j]
                    raise err
                diverged |= self.failed
                for j in range(m):
                    p[j] += (-stepsize / 2.) * tg[j]
            for j, o in enumerate(outs):
                H_new = obj[j] + p[j].size * half_log_2pi + np.dot(p[j], p[j]) / 2.
                k = 1. if H_old[j] > H_new else np.exp(H_old[j] - H_new)
                if not diverged[j] and np.isfinite(H_new) and uniforms[j][i] < k:
                    chains[j][i] = o.param_array[~o.fixed]
                    self.accepted[j] += 1
                else:
                    self.diverged[j] += int(diverged[j])
                    o.optimizer_array = x_old[j]
                    self._restore(j, *old[j])
