"""Generate tests/golden/*.npz by executing the REFERENCE's own arithmetic modules.

Run in the build container only (needs /root/reference):   python oracle/make_golden.py
The reference files are imported verbatim through oracle/ref_shim.py; nothing of them is
copied.  What the shim cannot import (paramz-based glue: GPy/core/gp.py,
GPyOpt/models/gpmodel*.py, multi_outputGP.py) is restated in `RefBackedModel` below from
gp.py:286-326,380-435, gaussian.py:94-111, normalizer.py:57-70, gpmodel.py:140-184 and
multi_outputGP.py:138-191 -- every number it returns is produced by reference code
(ExactGaussianInference / PosteriorExact / the reference kernels) plus the three scalar
conventions (+ymean, +noise, clip at 1e-10).

TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim as rs  # noqa: E402
from oracle import cpu_ref as R  # noqa: E402  (only the restated paramz Logexp transform is used by gen_hyper)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def ref_kernel(kind, d, variance, lengthscale, ARD):
    if kind == "se":
        return rs.ref_module("GPy.kern.src.se").SE(d, variance=variance, lengthscale=lengthscale, ARD=ARD)
    if kind == "rbf":
        return rs.ref_module("GPy.kern.src.rbf").RBF(d, variance=variance, lengthscale=lengthscale, ARD=ARD)
    st = rs.ref_module("GPy.kern.src.stationary")
    cls = {"matern52": st.Matern52, "matern32": st.Matern32}[kind]
    return cls(d, variance=variance, lengthscale=lengthscale, ARD=ARD)


class _Lik(object):
    """Stand-in for GPy.likelihoods.Gaussian (a paramz Parameterized): the two methods
    ExactGaussianInference.inference calls (gaussian.py:61-62 and :71-72)."""

    def __init__(self, variance):
        self.variance = variance

    def gaussian_variance(self, Y_metadata=None):
        return self.variance

    def exact_inference_gradients(self, dL_dKdiag, Y_metadata=None):
        return dL_dKdiag.sum()


class RefBackedGP(object):
    """One output: reference kernel + reference inference/posterior, fork conventions on top."""

    def __init__(self, kern, X, Y, noise_var):
        egi = rs.ref_module("GPy.inference.latent_function_inference.exact_gaussian_inference")
        self.kern, self.X, self.noise_var = kern, X, noise_var
        self.ymean = Y.mean(0)                      # normalizer.py:60-63 (std = 1)
        self.post, self.lml, self.grad_dict = egi.ExactGaussianInference().inference(kern, X, _Lik(noise_var), Y - self.ymean)

    def posterior_mean(self, Xn):                   # gp.py:393-399
        return self.post.raw_posterior_mean(self.kern, Xn, self.X) + self.ymean

    def posterior_variance(self, Xn):               # gp.py:416-417 + gpmodel.py:174
        return np.clip(self.post.raw_posterior_variance(self.kern, Xn, self.X) + self.noise_var, 1e-10, np.inf)

    def predict(self, Xn):                          # gp.py:314-324 + gpmodel.py:147
        mu, var = self.post._raw_predict(self.kern, Xn, self.X)
        return mu + self.ymean, np.clip(var + self.noise_var, 1e-10, np.inf)

    def posterior_mean_gradient(self, Xn):          # gp.py:438-461
        return self.kern.gradients_X(self.post.woodbury_vector.T, Xn, self.X)

    def posterior_variance_gradient(self, Xn):      # gp.py:464-490
        dv_dX = self.kern.gradients_X(np.eye(Xn.shape[0]), Xn)
        alpha = -2. * np.dot(self.kern.K(Xn, self.X), self.post.woodbury_inv)
        dv_dX += self.kern.gradients_X(alpha, Xn, self.X)
        return dv_dX


class RefBackedModel(object):
    """multi_outputGP.py:138-191 over RefBackedGP outputs."""
    analytical_gradient_prediction = True

    def __init__(self, gps, n_samples=10):
        self.output, self.output_dim, self.n_samples = gps, len(gps), n_samples

    def number_of_hyps_samples(self):
        return self.n_samples

    def set_hyperparameters(self, n):
        pass

    def predict(self, X):
        r = [g.predict(np.atleast_2d(X)) for g in self.output]
        return np.stack([a[:, 0] for a, _ in r]), np.stack([b[:, 0] for _, b in r])

    def posterior_mean(self, X):
        return np.stack([g.posterior_mean(X)[:, 0] for g in self.output])

    def posterior_variance(self, X):
        return np.stack([g.posterior_variance(X)[:, 0] for g in self.output])

    def posterior_mean_at_evaluated_points(self):
        return self.posterior_mean(self.output[0].X)

    def posterior_mean_gradient(self, X):           # multi_outputGP.py:284-294
        return np.stack([g.posterior_mean_gradient(X) for g in self.output])

    def posterior_variance_gradient(self, X):       # multi_outputGP.py:297-306
        return np.stack([g.posterior_variance_gradient(X) for g in self.output])


class CannedModel(object):
    """Duck-typed model with canned outputs (the reference's own Mock pattern,
    GPyOpt/testing/acquisitions_tests/test_ei_acquisition.py:11-26)."""
    analytical_gradient_prediction = True

    def __init__(self, mean, var, mu_eval, n_samples=10):
        self.mean, self.var, self.mu_eval = mean, var, mu_eval
        self.output_dim, self.n_samples = mean.shape[0], n_samples

    def number_of_hyps_samples(self):
        return self.n_samples

    def set_hyperparameters(self, n):
        pass

    def predict(self, X):
        return self.mean.copy(), self.var.copy()

    def posterior_mean(self, X):
        return self.mean.copy()

    def posterior_variance(self, X):
        return self.var.copy()

    def posterior_mean_at_evaluated_points(self):
        return self.mu_eval.copy()


# utilities as the experiment scripts define them (test_1a.py:89-92, test_1b.py:89-90,
# test_2a.py:60-62, test_3a.py:52-57, test_5a.py:48-52)
def U_neg_sq_dist(parameter, y):
    aux = (y.transpose() - parameter).transpose()
    return -np.sum(np.square(aux), axis=0)


def U_linear(parameter, y):
    return np.dot(parameter, y)


def U_neg_sum_exp(parameter, y):
    return np.sum(-np.exp(y), axis=0)


def make_U_neg_exp_cos(c):
    def U(parameter, y):
        y_copy = np.squeeze(y)
        aux = np.multiply(np.exp(-y_copy / np.pi), np.cos(np.pi * y_copy))
        return -np.dot(c, aux)
    return U


def make_U_rosenbrock(h):
    def U(a, y):
        val = 0
        for j in range(h):
            val -= (a - y[j]) ** 2 + 100 * y[j + h] ** 2
        return val
    return U


def ref_utility(func, support, prob, dfunc=None):
    ut = rs.ref_toplevel("utility")
    pd = rs.ref_toplevel("parameter_distribution")
    dist = pd.ParameterDistribution(continuous=False, support=support, prob_dist=prob)
    return ut.Utility(func=func, dfunc=dfunc, parameter_dist=dist, linear=False)


def dU_neg_sq_dist(parameter, y):                  # test_1a.py:94-96
    y_aux = np.squeeze(y)
    return -2 * (y_aux - parameter)


def dU_linear(parameter, y):
    return parameter


def dU_neg_sum_exp(parameter, y):                  # test_2a.py:64-65
    return -np.exp(y)


def gen_kernels():
    out = {}
    rng = np.random.RandomState(101)
    for (N, d, C, kinds) in [(40, 3, 30, ["rbf", "se", "matern52", "matern32"]), (96, 6, 100, ["rbf", "matern52"])]:
        X, Xs = rng.uniform(size=(N, d)), rng.uniform(size=(C, d))
        for kind in kinds:
            for ARD in (True, False):
                ls = rng.uniform(0.3, 1.2, size=d) if ARD else np.array([0.7])
                var = 1.7
                k = ref_kernel(kind, d, var, ls, ARD)
                tag = "%s_%s_N%d" % (kind, "ard" if ARD else "iso", N)
                out[tag + "_X"], out[tag + "_Xs"], out[tag + "_ls"], out[tag + "_var"] = X, Xs, ls, var
                out[tag + "_K"] = np.asarray(k.K(X))
                out[tag + "_Ks"] = np.asarray(k.K(X, Xs))
                out[tag + "_Kdiag"] = np.asarray(k.Kdiag(Xs))
    np.savez_compressed(os.path.join(OUT, "kernels.npz"), **out)


def gen_fit_predict():
    out = {}
    rng = np.random.RandomState(202)
    cases = [("se_N64", "se", 64, 2, False, 1e-6), ("se_N64_noisy", "se", 64, 2, False, 1e-2),
             ("rbf_N96", "rbf", 96, 6, True, 1e-6), ("m52_N96", "matern52", 96, 6, True, 1e-4)]
    for tag, kind, N, d, ARD, noise in cases:
        X, Xs = rng.uniform(size=(N, d)), rng.uniform(size=(150, d))
        a, b = rng.normal(size=d), rng.normal(size=d)
        Y = (np.sin(2 * np.pi * X.dot(a)) + 0.5 * np.cos(3 * X.dot(b)) + 0.3)[:, None]
        ls = 0.5 * np.sqrt(d) * (1 + 0.2 * rng.uniform(-1, 1, size=d)) if ARD else np.array([0.3])
        var = 2.0 if not ARD else 1.0
        gp = RefBackedGP(ref_kernel(kind, d, var, ls, ARD), X, Y, noise)
        for k_, v in dict(X=X, Xs=Xs, Y=Y, ls=ls, var=var, noise=noise, L=gp.post.woodbury_chol,
                          alpha=gp.post.woodbury_vector, lml=gp.lml,
                          raw_mean=gp.post.raw_posterior_mean(gp.kern, Xs, X),
                          raw_var=gp.post.raw_posterior_variance(gp.kern, Xs, X)).items():
            out[tag + "_" + k_] = np.asarray(v)
        mu2, var2 = gp.post._raw_predict(gp.kern, Xs, X)
        out[tag + "_rawpredict_mean"], out[tag + "_rawpredict_var"] = mu2, var2
    # jitter ladder (linalg.py:52-71; construction of GPy/testing/linalg_test.py:6-16)
    la = rs.ref_module("GPy.util.linalg")
    A = rng.randn(20, 100)
    A = A.dot(A.T)
    vals, vectors = np.linalg.eigh(A)
    vals[vals.argmin()] = 0
    default_jitter = 1e-6 * np.mean(vals)
    vals[vals.argmin()] = -default_jitter * (10 ** 3.5)
    A_corrupt = (vectors * vals).dot(vectors.T)
    L = la.jitchol(A_corrupt, maxtries=5)
    failed4 = False
    try:
        la.jitchol(A_corrupt, maxtries=4)
    except Exception:
        failed4 = True
    out["jit_A"], out["jit_L"], out["jit_failed_with_4"] = A_corrupt, L, np.array(failed4)
    np.savez_compressed(os.path.join(OUT, "fit_predict.npz"), **out)


def gen_acq_canned():
    out = {}
    rng = np.random.RandomState(303)
    m, C, Nn = 3, 60, 25
    mean = rng.normal(size=(m, C))
    var = rng.uniform(1e-8, 0.5, size=(m, C))
    var[:, :4] = 1e-10                       # clipped-variance candidates
    var[:, 4] = 0.0                          # sigma < 1e-10 branch (maEI.py:155-156)
    mu_eval = rng.normal(size=(m, Nn))
    support = rng.uniform(0.1, 1.0, size=(2, m))
    prob = np.array([0.3, 0.7])
    out.update(mean=mean, var=var, mu_eval=mu_eval, support=support, prob=prob)
    model = CannedModel(mean, var, mu_eval)
    Xd = np.zeros((C, 2))
    for name in ("maEI", "maPI"):
        cls = getattr(rs.ref_toplevel(name), name)
        acq = cls(model, None, optimizer=None, utility=ref_utility(U_linear, support, prob))
        out[name + "_full"] = acq._compute_acq(Xd)
        acq1 = cls(model, None, optimizer=None, utility=ref_utility(U_linear, support[:1], np.ones(1)))
        out[name + "_L1"] = acq1._compute_acq(Xd)
    # not-full-support path: >= 20 support points -> theta sampled from the global RNG
    big_support = rng.uniform(0.1, 1.0, size=(24, m))
    big_prob = np.full(24, 1.0 / 24)
    out.update(big_support=big_support, big_prob=big_prob)
    acq = rs.ref_toplevel("maEI").maEI(model, None, optimizer=None, utility=ref_utility(U_linear, big_support, big_prob))
    np.random.seed(77)
    out["maEI_sampled_seed77"] = acq._compute_acq(Xd)
    # single-output EI.py / PI.py twins
    model1 = CannedModel(mean[:1], var[:1], mu_eval[:1])
    for name in ("EI", "PI"):
        cls = getattr(rs.ref_toplevel(name), name)
        a1 = cls(model1, None, optimizer=None, utility=ref_utility(U_linear, np.array([[1.0]]), np.ones(1)))
        out[name + "_single"] = a1._compute_acq(Xd)
    # Monte-Carlo acquisitions
    sig_model = CannedModel(mean, np.clip(var, 1e-10, np.inf), mu_eval)
    thetas = rng.normal(size=(2, m))
    out["mc_thetas"] = thetas
    utils = {"neg_sq_dist": U_neg_sq_dist, "neg_sum_exp": U_neg_sum_exp,
             "neg_exp_cos": make_U_neg_exp_cos(np.array([1.0, 2.0, 5.0]))}
    for name in ("uEI_noiseless", "uPI"):
        cls = getattr(rs.ref_toplevel(name), name)
        for uname, U in utils.items():
            if uname == "neg_sq_dist":
                sup, pr = thetas, prob
            else:
                sup, pr = np.ones((1, 1)), np.ones(1)      # test_2a.py:54-56: parameter unused
            np.random.seed(5)
            acq = cls(sig_model, None, optimizer=None, utility=ref_utility(U, sup, pr))
            out["mc_W25"] = acq.W_samples
            out["%s_%s_seq" % (name, uname)] = acq._compute_acq(Xd, parallel=False)
    # rosenbrock-like utility of test_5a.py (m = 2(d-1) = 4 outputs, scalar parameter a = 1)
    m4 = 4
    mean4, var4, mu_eval4 = rng.normal(size=(m4, 30)) * 0.5, rng.uniform(1e-6, 0.2, size=(m4, 30)), rng.normal(size=(m4, 12)) * 0.5
    out.update(rb_mean=mean4, rb_var=var4, rb_mu_eval=mu_eval4)
    np.random.seed(6)
    acq = rs.ref_toplevel("uEI_noiseless").uEI_noiseless(
        CannedModel(mean4, var4, mu_eval4), None, optimizer=None,
        utility=ref_utility(make_U_rosenbrock(2), np.atleast_1d([1.0]), np.ones(1)))
    out["rb_W"] = acq.W_samples
    out["rb_uEI"] = acq._compute_acq(np.zeros((30, 3)), parallel=False)
    np.savez_compressed(os.path.join(OUT, "acq_canned.npz"), **out)


def gen_e2e():
    """Whole path on config-1 shape (N=64,d=2,m=1,C=400,S=25, SE kernel, -(y-theta)^2) and a small
    config-2 shape (m=4 RBF ARD, N=128, d=6, S=32, C=256): reference inference + posterior +
    acquisition classes; selection as anchor_points_generator.py:59-61."""
    from oracle import cpu_ref
    out = {}
    for tag, kind, N, d, m, C, S, seed in [("cfg1", "se", 64, 2, 1, 400, 25, 1235), ("cfg2s", "rbf", 128, 6, 4, 256, 32, 1236),
                                           ("cfg2", "rbf", 1024, 6, 4, 8192, 256, 1236)]:   # BASELINE configs[1] at full size
        p = cpu_ref.synthetic_problem(N, d, m, C, S, seed)
        gps = [RefBackedGP(ref_kernel(kind, d, p["variances"][j], p["lengthscales"][j], True), p["X"], p["Y"][j], p["noise"][j])
               for j in range(m)]
        model = RefBackedModel(gps)
        theta = np.array([[0.2 * (j + 1) for j in range(m)]])
        np.random.seed(11)
        uei = rs.ref_toplevel("uEI_noiseless").uEI_noiseless(
            model, None, optimizer=None, utility=ref_utility(U_neg_sq_dist, theta, np.ones(1)))
        uei.W_samples = p["W"]
        a_uei = uei._compute_acq(p["Xc"], parallel=False)
        if tag == "cfg1":   # the pathos variant (uEI_noiseless.py:85-116) computes the same numbers
            out[tag + "_uEI_par"] = uei._compute_acq(p["Xc"], parallel=True)
        if tag == "cfg2":      # full-size selection vector (SURVEY 8c G-SEL): keep the file small
            a_maei_full = rs.ref_toplevel("maEI").maEI(model, None, optimizer=None, utility=ref_utility(
                U_linear, np.full((1, m), 1.0 / m), np.ones(1)))._compute_acq(p["Xc"])
            out.update({tag + "_seed": seed, tag + "_theta": theta, tag + "_uEI": a_uei, tag + "_maEI": a_maei_full,
                        tag + "_sel_uEI": np.argsort((-a_uei).flatten())[:16], tag + "_sel_maEI": np.argsort((-a_maei_full).flatten())[:16],
                        tag + "_lml": np.array([g.lml for g in gps])})
            continue
        upi = rs.ref_toplevel("uPI").uPI(model, None, optimizer=None, utility=ref_utility(U_neg_sq_dist, theta, np.ones(1)))
        upi.W_samples = p["W"]
        a_upi = upi._compute_acq(p["Xc"], parallel=False)
        th_lin = np.full((1, m), 1.0 / m)
        maei = rs.ref_toplevel("maEI").maEI(model, None, optimizer=None, utility=ref_utility(U_linear, th_lin, np.ones(1)))
        a_maei = maei._compute_acq(p["Xc"])
        mean, var = model.predict(p["Xc"])
        out.update({tag + "_seed": seed, tag + "_theta": theta, tag + "_theta_lin": th_lin,
                    tag + "_uEI": a_uei, tag + "_uPI": a_upi, tag + "_maEI": a_maei,
                    tag + "_mean": mean, tag + "_var": var,
                    tag + "_post_mean": model.posterior_mean(p["Xc"]), tag + "_post_var": model.posterior_variance(p["Xc"]),
                    tag + "_mu_eval": model.posterior_mean_at_evaluated_points(),
                    tag + "_lml": np.array([g.lml for g in gps]),
                    tag + "_sel_uEI": np.argsort((-a_uei).flatten())[:16],
                    tag + "_sel_maEI": np.argsort((-a_maei).flatten())[:16]})
    np.savez_compressed(os.path.join(OUT, "e2e.npz"), **out)


def gen_gradients():
    """Gradient rows: kern.gradients_X, posterior mean/variance gradients and the
    acquisitions' _compute_acq_withGradients, all computed by the reference's code."""
    from oracle import cpu_ref
    out = {}
    for tag, kind, N, d, m, n, S, seed, noise in [("se", "se", 60, 3, 2, 7, 6, 31, 1e-4), ("rbf", "rbf", 60, 3, 2, 7, 6, 32, 1e-4),
                                                   ("m52", "matern52", 50, 4, 3, 5, 5, 33, 1e-3)]:
        p = cpu_ref.synthetic_problem(N, d, m, n, S, seed, noise=noise)
        gps = [RefBackedGP(ref_kernel(kind, d, p["variances"][j], p["lengthscales"][j], True), p["X"], p["Y"][j], p["noise"][j])
               for j in range(m)]
        model = RefBackedModel(gps)
        Xc = p["Xc"]
        out[tag + "_seed"], out[tag + "_noise"] = seed, noise
        out[tag + "_dmean"] = model.posterior_mean_gradient(Xc)
        out[tag + "_dvar"] = model.posterior_variance_gradient(Xc)
        rng = np.random.RandomState(seed)
        dL = rng.normal(size=(n, N))
        out[tag + "_dL"] = dL
        out[tag + "_gradX"] = gps[0].kern.gradients_X(dL, Xc, p["X"])
        support, prob = rng.normal(size=(2, m)) * 0.5, np.array([0.4, 0.6])
        out[tag + "_support"], out[tag + "_prob"] = support, prob
        for name in ("maEI", "maPI"):
            acq = getattr(rs.ref_toplevel(name), name)(model, None, optimizer=None, utility=ref_utility(U_linear, support, prob, dU_linear))
            a, da = acq._compute_acq_withGradients(Xc)
            out["%s_%s_acq" % (tag, name)], out["%s_%s_dacq" % (tag, name)] = a, da
        for uname, U, dU in (("neg_sq_dist", U_neg_sq_dist, dU_neg_sq_dist), ("neg_sum_exp", U_neg_sum_exp, dU_neg_sum_exp)):
            sup, pr = (support, prob) if uname == "neg_sq_dist" else (np.ones((1, 1)), np.ones(1))
            np.random.seed(3)
            acq = rs.ref_toplevel("uEI_noiseless").uEI_noiseless(model, None, optimizer=None, utility=ref_utility(U, sup, pr, dU))
            acq.W_samples = p["W"]
            a, da = acq._compute_acq_withGradients(Xc)
            out["%s_uEI_%s_acq" % (tag, uname)], out["%s_uEI_%s_dacq" % (tag, uname)] = a, da
    np.savez_compressed(os.path.join(OUT, "gradients.npz"), **out)


def gen_hypergrads():
    """Rank-3 building block: d log-marginal / d (variance, lengthscale, noise) from the reference's
    exact_gaussian_inference.py:61-63 + kern.update_gradients_full (gp.py:258)."""
    out = {}
    rng = np.random.RandomState(404)
    for tag, kind, N, d, ARD, noise in [("se_ard", "se", 70, 3, True, 1e-3), ("se_iso", "se", 70, 3, False, 1e-3),
                                        ("rbf_ard", "rbf", 70, 3, True, 1e-2), ("rbf_iso", "rbf", 50, 2, False, 1e-2),
                                        ("m52_ard", "matern52", 60, 4, True, 1e-3), ("m32_iso", "matern32", 60, 4, False, 1e-3)]:
        X = rng.uniform(size=(N, d))
        Y = (np.sin(3 * X.sum(1)) + 0.1 * rng.normal(size=N))[:, None]
        ls = rng.uniform(0.3, 0.9, size=d) if ARD else np.array([0.5])
        var = 1.3
        gp = RefBackedGP(ref_kernel(kind, d, var, ls, ARD), X, Y, noise)
        gp.kern.update_gradients_full(gp.grad_dict["dL_dK"], X)
        out.update({tag + "_X": X, tag + "_Y": Y, tag + "_ls": ls, tag + "_var": var, tag + "_noise": noise,
                    tag + "_lml": gp.lml, tag + "_dvar": np.asarray(gp.kern.variance.gradient, dtype=float),
                    tag + "_dls": np.atleast_1d(np.asarray(gp.kern.lengthscale.gradient, dtype=float)),
                    tag + "_dnoise": np.asarray(gp.grad_dict["dL_dthetaL"], dtype=float)})
    np.savez_compressed(os.path.join(OUT, "hypergrads.npz"), **out)


class RefDuckModel(object):
    """What hmc.py:21-69 needs from a GPy model, assembled from the reference's own pieces: kernel + hyper-gradients
    (kern.update_gradients_full), ExactGaussianInference (log-marginal, dL_dK, dL_dthetaL), priors.Gamma.  Only the
    paramz glue is restated: Logexp transform (cpu_ref.logexp_*), parameter ordering [variance, lengthscale, noise],
    fixed-noise handling (constrain_fixed)."""

    def __init__(self, kind, X, Y, variance, lengthscale, ARD, noise_var, noise_fixed):
        pri = rs.ref_module("GPy.core.parameterization.priors")
        self.prior = pri.Gamma.from_EV(2., 4.)            # gpmodel.py:67-68
        self.kind, self.X, self.Y, self.ARD = kind, X, Y, ARD
        self.param_array = np.concatenate(([variance], np.atleast_1d(lengthscale), [noise_var])).astype(float)
        self.fixed = np.zeros(self.param_array.size, dtype=bool)
        self.fixed[-1] = noise_fixed
        self._key = None

    def _gp(self):
        key = self.param_array.tobytes()
        if key != self._key:
            p = self.param_array
            self._gp_obj = RefBackedGP(ref_kernel(self.kind, self.X.shape[1], p[0], p[1:-1].copy(), self.ARD), self.X, self.Y, p[-1])
            self._gp_obj.kern.update_gradients_full(self._gp_obj.grad_dict["dL_dK"], self.X)
            self._key = key
        return self._gp_obj

    @property
    def unfixed_param_array(self):
        return self.param_array[~self.fixed].copy()

    @property
    def optimizer_array(self):
        return R.logexp_finv(self.param_array[~self.fixed])

    @optimizer_array.setter
    def optimizer_array(self, x):
        self.param_array[~self.fixed] = R.logexp_f(x)

    def log_prior(self):                                  # priorizable.py:49-65
        x = self.param_array
        return float(self.prior.lnpdf(x).sum() + R.logexp_log_jacobian(x[~self.fixed]).sum())

    def objective_function(self):                         # model.py:72-84
        return -float(self._gp().lml) - self.log_prior()

    def objective_function_gradients(self):               # model.py:86-104
        gp = self._gp()
        dl = np.concatenate((np.atleast_1d(np.asarray(gp.kern.variance.gradient, dtype=float)),
                             np.atleast_1d(np.asarray(gp.kern.lengthscale.gradient, dtype=float)),
                             np.atleast_1d(np.asarray(gp.grad_dict["dL_dthetaL"], dtype=float))))
        dp = self.prior.lnpdf_grad(self.param_array)
        dp[~self.fixed] += R.logexp_log_jacobian_grad(self.param_array[~self.fixed])
        return -(dl + dp)

    def _transform_gradients(self, g):
        free = ~self.fixed
        return R.logexp_gradfactor(self.param_array[free], np.asarray(g, dtype=float)[free])


def gen_hyper():
    """Rank 3: objective (= -log-marginal - log-prior incl. the Logexp Jacobian), its gradients, and seeded HMC chains
    produced by the reference's GPy/inference/mcmc/hmc.py (executed verbatim) over RefDuckModel."""
    hmc = rs.ref_module("GPy.inference.mcmc.hmc")
    out = {}
    rng = np.random.RandomState(505)
    for tag, kind, N, d, ARD, noise, fixed in [("se_ard_fixed", "se", 30, 2, True, 1e-6, True), ("rbf_iso_free", "rbf", 36, 3, False, 0.02, False),
                                               ("m52_ard_free", "matern52", 40, 3, True, 0.01, False)]:
        X = rng.uniform(size=(N, d))
        Y = (np.sin(3 * X.sum(1)) + np.cos(2 * X[:, 0]) + (0.0 if fixed else 0.05) * rng.normal(size=N))[:, None]
        ls = rng.uniform(0.4, 1.2, size=d) if ARD else np.array([0.7])
        var = 1.2
        model = RefDuckModel(kind, X, Y, var, ls, ARD, noise, fixed)
        out.update({tag + "_X": X, tag + "_Y": Y, tag + "_theta0": model.param_array.copy(), tag + "_fixed": model.fixed.copy(),
                    tag + "_obj": model.objective_function(), tag + "_grad": model.objective_function_gradients(),
                    tag + "_tgrad": model._transform_gradients(model.objective_function_gradients()), tag + "_x0": model.optimizer_array})
        seed, ns, iters, step = 9000 + N, 14, 6, (0.01 if fixed else 0.05)
        np.random.seed(seed)
        chain = hmc.HMC(model, stepsize=step).sample(num_samples=ns, hmc_iters=iters)
        out.update({tag + "_seed": seed, tag + "_num_samples": ns, tag + "_hmc_iters": iters, tag + "_stepsize": step, tag + "_chain": chain,
                    tag + "_theta_end": model.param_array.copy()})
    np.savez_compressed(os.path.join(OUT, "hyper.npz"), **out)


def gen_hyper_defaults():
    """Rank 3 at the reference's DEFAULT sampler settings (GPyOpt/models/gpmodel.py:31: step_size = 1e-1, leapfrog_steps = 20)
    with exact_feval (noise fixed at 1e-6, :72-73) and the default kernel SE(variance=1, ARD) (:58): the flow of
    GPModel.updateModel (:102-119) -- optimise, 1 % jitter of the parameter vector, HMC -- with hmc.py executed verbatim.
    Records the chain (or the draw at which jitchol's LinAlgError left updateModel), so that the device path can be held to
    the reference's behaviour at these settings: acceptance, rejection, or the exception."""
    hmc = rs.ref_module("GPy.inference.mcmc.hmc")
    out = {}
    for tag, N, d, ns, seed in [("N64", 64, 2, 40, 6401), ("N256", 256, 4, 24, 25601)]:
        rng = np.random.RandomState(700 + N)
        X = rng.uniform(size=(N, d))
        Y = (np.sin(3 * X.sum(1)) + np.cos(2 * X[:, 0]) * X[:, 1])[:, None]
        model = RefDuckModel("se", X, Y, 1.0, np.ones(d), True, 1e-6, True)
        R.optimize_hyper(model, 200)                              # gpmodel.py:115 (paramz opt_lbfgsb through SciPy's L-BFGS-B)
        theta_opt = model.param_array.copy()
        np.random.seed(seed)
        model.param_array[:] = model.param_array * (1. + np.random.randn(model.param_array.size) * 0.01)    # :116
        theta_start = model.param_array.copy()
        sampler = hmc.HMC(model, stepsize=1e-1)
        chain, raised_at, objs = np.full((ns, int(np.sum(~model.fixed))), np.nan), -1, []
        # hmc.py:44-59 draw by draw (sample(1) per draw consumes the RNG exactly like sample(ns): momentum, then rand)
        for i in range(ns):
            try:
                chain[i] = sampler.sample(num_samples=1, hmc_iters=20)[0]
                objs.append(model.objective_function())
            except Exception as e:                                # scipy / numpy LinAlgError out of jitchol (linalg.py:71)
                raised_at = i
                out[tag + "_raised"] = np.array(type(e).__name__ + ": " + str(e))
                break
        acc = int(np.sum(np.any(np.diff(np.vstack([theta_start[~model.fixed], chain[:len(objs)]]), axis=0) != 0, axis=1)))
        print("hyper defaults %s: %d draws, accepted %d, raised_at %d, theta_opt %s" % (tag, len(objs), acc, raised_at, theta_opt))
        out.update({tag + "_X": X, tag + "_Y": Y, tag + "_theta_opt": theta_opt, tag + "_theta_start": theta_start, tag + "_seed": seed,
                    tag + "_chain": chain, tag + "_raised_at": raised_at, tag + "_objective": np.array(objs),
                    tag + "_theta_end": model.param_array.copy(), tag + "_num_samples": ns})
    np.savez_compressed(os.path.join(OUT, "hyper_defaults.npz"), **out)


def gen_dense():
    """BASELINE configs[1] at FULL size (m=4 RBF ARD, N=1024, d=6, S=256, C=8192) on a workload whose acquisition is DENSE
    (VERDICT r2 item 1b): the synthetic targets are scaled by 1e-3, so the spread of the posterior mean over the design
    is of the order of the posterior standard deviation (1e-3 ... 1e-2) and most candidates have a non-zero EI / PI -- on the
    unscaled targets ONE of the 8192 candidates has (uEI_noiseless.py:63-83 gives exact zeros elsewhere).  The reference's own
    uEI_noiseless / uPI (theta fixed at 0.2 (j + 1): the utility is ~linear in the small outputs), maEI / maPI classes on the
    reference's inference + posterior."""
    from oracle import cpu_ref
    tag, kind, N, d, m, C, S, seed, amp = "cfg2d", "rbf", 1024, 6, 4, 8192, 256, 1236, 1e-3
    p = cpu_ref.synthetic_problem(N, d, m, C, S, seed)
    Ys = [amp * y for y in p["Y"]]
    gps = [RefBackedGP(ref_kernel(kind, d, p["variances"][j], p["lengthscales"][j], True), p["X"], Ys[j], p["noise"][j]) for j in range(m)]
    model = RefBackedModel(gps, n_samples=1)     # (fixed hyper-parameters: the reference's min(10, .) passes are identical, maEI.py:35,85)
    theta = np.array([[0.2 * (j + 1) for j in range(m)]])
    th_lin = np.full((1, m), 1.0 / m)
    out = {tag + "_seed": seed, tag + "_amp": amp, tag + "_theta": theta, tag + "_theta_lin": th_lin}
    np.random.seed(11)
    for name, mod, util in [("uEI", "uEI_noiseless", ref_utility(U_neg_sq_dist, theta, np.ones(1))),
                            ("uPI", "uPI", ref_utility(U_neg_sq_dist, theta, np.ones(1))),
                            ("maEI", "maEI", ref_utility(U_linear, th_lin, np.ones(1))),
                            ("maPI", "maPI", ref_utility(U_linear, th_lin, np.ones(1)))]:
        acq = getattr(rs.ref_toplevel(mod), mod)(model, None, optimizer=None, utility=util)
        if name in ("uEI", "uPI"):
            acq.W_samples = p["W"]
            a = acq._compute_acq(p["Xc"], parallel=False)      # (parallel=True is the pathos map over single candidates: same numbers, hours)
        else:
            a = acq._compute_acq(p["Xc"])
        a = np.asarray(a, dtype=float).reshape(C, 1)
        print("dense %s: max %.3e, fraction above 1e-3 max: %.3f" % (name, a.max(), float((a > 1e-3 * a.max()).mean())))
        out[tag + "_" + name] = a
        out[tag + "_sel_" + name] = np.argsort((-a).flatten(), kind="stable")[:16]
    np.savez_compressed(os.path.join(OUT, "dense.npz"), **out)


def gen_fullcov():
    """multi_outputGP.predict(X, full_cov=True) (multi_outputGP.py:138-149): the reference's PosteriorExact._raw_predict with
    full_cov=True (posterior.py:274-283), + noise on the diagonal (gaussian.py:95-97), clipped entry-wise at 1e-10
    (gpmodel_fixed_hyps.py:84-86), and the wrapper's column 0 (multi_outputGP.py:146-148)."""
    from oracle import cpu_ref
    out = {}
    for tag, kinds, N, d, C, seed in [("rbf", ["rbf"] * 3, 150, 3, 40, 77), ("mixed", ["matern52", "rbf", "matern32", "se"], 90, 2, 25, 78)]:
        m = len(kinds)
        p = cpu_ref.synthetic_problem(N, d, m, C, 4, seed)
        Xc = p["Xc"].copy()
        Xc[1] = Xc[0] + 1e-3                        # a neighbour of point 0 (covariance close to the variance)
        Xc[2] = p["X"][5]                           # a training input (posterior covariance ~ 0 or slightly negative: clipped)
        cov, mean = np.empty((m, C)), np.empty((m, C))
        for j in range(m):
            g = RefBackedGP(ref_kernel(kinds[j], d, p["variances"][j], p["lengthscales"][j], True), p["X"], p["Y"][j], p["noise"][j])
            mu, var = g.post._raw_predict(g.kern, Xc, g.X, full_cov=True)
            var = var + np.eye(C) * g.noise_var                              # gaussian.py:97
            var = np.clip(var, 1e-10, np.inf)                                # gpmodel_fixed_hyps.py:86
            mean[j], cov[j] = (mu + g.ymean)[:, 0], var[:, 0]                # multi_outputGP.py:146-148
        out.update({tag + "_seed": seed, tag + "_N": N, tag + "_d": d, tag + "_C": C, tag + "_Xc": Xc, tag + "_mean": mean, tag + "_cov0": cov,
                    tag + "_kinds": np.array(kinds)})
        print("fullcov %s: clipped entries %d of %d" % (tag, int((cov == 1e-10).sum()), cov.size))
    np.savez_compressed(os.path.join(OUT, "fullcov.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:                         # e.g. `python oracle/make_golden.py gen_fullcov`: one generator only
        for name in sys.argv[1:]:
            globals()[name]()
        raise SystemExit(0)
    if not rs.available():
        raise SystemExit("reference tree not mounted; golden vectors can only be generated in the build container")
    os.makedirs(OUT, exist_ok=True)
    gen_kernels()
    gen_fit_predict()
    gen_acq_canned()
    gen_e2e()
    gen_gradients()
    gen_hypergrads()
    gen_hyper()
    gen_hyper_defaults()
    gen_dense()
    gen_fullcov()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
