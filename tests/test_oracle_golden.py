"""The oracle (oracle/cpu_ref.py) against golden vectors produced by the reference's own code
(oracle/make_golden.py) and against the reference's known-answer tests.  CPU only."""
import numpy as np
import pytest
import scipy.linalg

from oracle import cpu_ref as R

KINDS = ["rbf", "se", "matern52", "matern32"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("ard", ["ard", "iso"])
def test_kernels_small(golden, kind, ard):
    g = golden("kernels")
    t = "%s_%s_N40" % (kind, ard)
    K = R.kern_K(kind, g[t + "_X"], None, g[t + "_var"], g[t + "_ls"])
    Ks = R.kern_K(kind, g[t + "_X"], g[t + "_Xs"], g[t + "_var"], g[t + "_ls"])
    np.testing.assert_allclose(K, g[t + "_K"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(Ks, g[t + "_Ks"], rtol=1e-13, atol=1e-15)
    np.testing.assert_array_equal(R.kern_Kdiag(g[t + "_Xs"], g[t + "_var"]), g[t + "_Kdiag"])
    # direct-difference form (what the device computes) == expansion form to round-off
    Kd = R.kern_K_direct(kind, g[t + "_X"], g[t + "_Xs"], g[t + "_var"], g[t + "_ls"])
    np.testing.assert_allclose(Kd, g[t + "_Ks"], rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
@pytest.mark.parametrize("ard", ["ard", "iso"])
def test_kernels_n96(golden, kind, ard):
    g = golden("kernels")
    t = "%s_%s_N96" % (kind, ard)
    np.testing.assert_allclose(R.kern_K(kind, g[t + "_X"], None, g[t + "_var"], g[t + "_ls"]), g[t + "_K"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(R.kern_K(kind, g[t + "_X"], g[t + "_Xs"], g[t + "_var"], g[t + "_ls"]), g[t + "_Ks"], rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("tag,kind", [("se_N64", "se"), ("se_N64_noisy", "se"), ("rbf_N96", "rbf"), ("m52_N96", "matern52")])
def test_fit_predict(golden, tag, kind):
    g = golden("fit_predict")
    fit = R.GPFit(kind, g[tag + "_X"], g[tag + "_Y"], g[tag + "_var"], g[tag + "_ls"], g[tag + "_noise"])
    assert fit.jitter == 0.0
    np.testing.assert_allclose(fit.L, g[tag + "_L"], rtol=1e-9, atol=1e-12)
    scale = np.abs(g[tag + "_alpha"]).max()
    np.testing.assert_allclose(fit.alpha, g[tag + "_alpha"], rtol=1e-7, atol=1e-9 * scale)
    np.testing.assert_allclose(fit.log_marginal, g[tag + "_lml"], rtol=1e-10)
    Xs = g[tag + "_Xs"]
    np.testing.assert_allclose(fit.raw_posterior_mean(Xs), g[tag + "_raw_mean"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(fit.raw_posterior_variance(Xs), g[tag + "_raw_var"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(g[tag + "_rawpredict_mean"], g[tag + "_raw_mean"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(g[tag + "_rawpredict_var"], g[tag + "_raw_var"], rtol=1e-12, atol=0)


def test_jitter_ladder(golden):
    """linalg.py:52-71 / GPy/testing/linalg_test.py:18-37: 5 rounds succeed, 4 fail."""
    g = golden("fit_predict")
    assert bool(g["jit_failed_with_4"])
    L, jitter = R.jitchol(g["jit_A"], maxtries=5)
    np.testing.assert_allclose(L, g["jit_L"], rtol=1e-7, atol=1e-9)
    assert jitter == pytest.approx(np.diag(g["jit_A"]).mean() * 1e-6 * 1e4)
    diff = L.dot(L.T) - g["jit_A"]
    np.testing.assert_allclose(diff, np.eye(L.shape[0]) * np.diag(diff).mean(), atol=1e-10)
    with pytest.raises(scipy.linalg.LinAlgError):
        R.jitchol(g["jit_A"], maxtries=4)
    bad = g["jit_A"].copy()
    bad[3, 3] = -1.0
    with pytest.raises(scipy.linalg.LinAlgError):
        R.jitchol(bad)


def test_raw_predict_known_answer():
    """GPy/testing/model_tests.py:63-82 re-derived: predict_noiseless == explicit K* (K+s2 I)^-1 algebra
    on the mean-centred targets."""
    rng = np.random.RandomState(0)
    N, Nn = 20, 50
    X = rng.uniform(-3.0, 3.0, (N, 1))
    Y = np.sin(X) + rng.randn(N, 1) * 0.05
    Xn = rng.uniform(-3.0, 3.0, (Nn, 1))
    var, ls, noise = 1.3, 0.8, 0.5
    fit = R.GPFit("rbf", X, Y, var, ls, noise)
    K = R.kern_K("rbf", X, None, var, ls)
    Kinv = np.linalg.pinv(K + np.eye(N) * (noise + 1e-8))
    Kx = R.kern_K("rbf", Xn, X, var, ls)
    K_hat = R.kern_K("rbf", Xn, None, var, ls) - Kx.dot(Kinv).dot(Kx.T)
    mu_hat = Kx.dot(Kinv).dot(Y - Y.mean())
    mu, v = fit.predict_noiseless(Xn)
    np.testing.assert_almost_equal(mu - Y.mean(), mu_hat)
    np.testing.assert_almost_equal(np.clip(np.diag(K_hat)[:, None], 1e-10, np.inf), v)


def test_ei_known_answer():
    """GPyOpt/testing/acquisitions_tests/test_ei_acquisition.py:18-37: EI(m=1, s=3, fmin=0.1,
    jitter 0.01) = 0.79646919 under the stock minimisation convention u=(fmin-m-jitter)/s
    (GPyOpt/util/general.py:113-128).  Mirrored through our maximisation quantiles:
    u_max(fmax=-fmin+jitter... ) -> use m' = -m, fmax = -fmin + jitter."""
    m, s, fmin, jit = 1.0, 3.0, 0.1, 0.01
    phi, Phi, u = R.get_quantiles(-(fmin - jit), np.array([-m]), np.array([s]))
    assert s * (u * Phi + phi) == pytest.approx(0.79646919, abs=1e-8)


def test_acq_canned(golden):
    g = golden("acq_canned")
    mean, var, mu_eval = g["mean"], g["var"], g["mu_eval"]
    for name, kind in (("maEI", "EI"), ("maPI", "PI")):
        a, _ = R.ma_acq(mean, var, mu_eval, g["support"], g["prob"], kind)
        np.testing.assert_allclose(a, g[name + "_full"], rtol=1e-12, atol=1e-300)
        a, _ = R.ma_acq(mean, var, mu_eval, g["support"][:1], np.ones(1), kind)
        np.testing.assert_allclose(a, g[name + "_L1"], rtol=1e-12, atol=1e-300)
    for name, kind in (("EI", "EI"), ("PI", "PI")):
        a, _ = R.ma_acq(mean[:1], var[:1], mu_eval[:1], np.array([[1.0]]), np.ones(1), kind)
        np.testing.assert_allclose(a, g[name + "_single"], rtol=1e-12, atol=1e-300)
    # sampled-theta path: maEI.py:46 draws 3 thetas with np.random.choice (parameter_distribution.py:27)
    np.random.seed(77)
    idx = np.random.choice(24, size=3, p=g["big_prob"])
    a, _ = R.ma_acq(mean, var, mu_eval, g["big_support"][idx], None, "EI")
    np.testing.assert_allclose(a, g["maEI_sampled_seed77"], rtol=1e-12, atol=1e-300)
    sigma = np.sqrt(np.clip(var, 1e-10, np.inf))
    W = g["mc_W25"]
    for name, kind in (("uEI_noiseless", "EI"), ("uPI", "PI")):
        a, _ = R.mc_acq(mean, sigma, mu_eval, W, "neg_sq_dist", g["mc_thetas"], g["prob"], kind)
        np.testing.assert_allclose(a, g[name + "_neg_sq_dist_seq"], rtol=1e-11, atol=1e-14)
        a2, _ = R.mc_acq_loop(mean, sigma, mu_eval, W, "neg_sq_dist", g["mc_thetas"], g["prob"], kind)
        np.testing.assert_allclose(a2, g[name + "_neg_sq_dist_seq"], rtol=1e-13, atol=1e-15)
        a, _ = R.mc_acq(mean, sigma, mu_eval, W, "neg_sum_exp", np.ones((1, 1)), np.ones(1), kind)
        np.testing.assert_allclose(a, g[name + "_neg_sum_exp_seq"], rtol=1e-11, atol=1e-14)
        a, _ = R.mc_acq(mean, sigma, mu_eval, W, "neg_exp_cos", np.ones((1, 1)), np.ones(1), kind,
                        util_params=np.array([1.0, 2.0, 5.0]))
        np.testing.assert_allclose(a, g[name + "_neg_exp_cos_seq"], rtol=1e-11, atol=1e-14)
    a, _ = R.mc_acq(g["rb_mean"], np.sqrt(g["rb_var"]), g["rb_mu_eval"], g["rb_W"], "rosenbrock",
                    np.array([[1.0]]), np.ones(1), "EI")
    np.testing.assert_allclose(a, g["rb_uEI"], rtol=1e-11, atol=1e-14)


def _check_selection(acq, order, ref_order):
    """Top-k indices must match the reference's np.argsort(-acq)[:k] wherever the acquisition
    value is untied; among exact ties (typically acq == 0) numpy's default quicksort order is
    unspecified, ours is lowest-index-first, so only the VALUES must agree there."""
    acq = np.asarray(acq).flatten()
    np.testing.assert_array_equal(acq[order], acq[ref_order])
    for k, (i, j) in enumerate(zip(order, ref_order)):
        if np.sum(acq == acq[j]) == 1:
            assert i == j, (k, i, j)
    assert order[0] == ref_order[0]


@pytest.mark.parametrize("tag,kind,N,d,m,C,S", [("cfg1", "se", 64, 2, 1, 400, 25), ("cfg2s", "rbf", 128, 6, 4, 256, 32)])
def test_e2e(golden, tag, kind, N, d, m, C, S):
    g = golden("e2e")
    p = R.synthetic_problem(N, d, m, C, S, int(g[tag + "_seed"]))
    model = R.MultiOutputGPRef(kind, p["variances"], p["lengthscales"], p["noise"])
    model.updateModel(p["X"], p["Y"])
    np.testing.assert_allclose([o.log_marginal for o in model.output], g[tag + "_lml"], rtol=1e-9)
    mean, var = model.predict(p["Xc"])
    sf2 = 1.0
    np.testing.assert_allclose(mean, g[tag + "_mean"], rtol=1e-6, atol=1e-7)
    assert np.abs(var - g[tag + "_var"]).max() <= 1e-9 * sf2
    np.testing.assert_allclose(model.posterior_mean(p["Xc"]), g[tag + "_post_mean"], rtol=1e-6, atol=1e-7)
    assert np.abs(model.posterior_variance(p["Xc"]) - g[tag + "_post_var"]).max() <= 1e-9 * sf2
    np.testing.assert_allclose(model.posterior_mean_at_evaluated_points(), g[tag + "_mu_eval"], rtol=1e-6, atol=1e-7)
    a, order, best = R.batch_uEI(model, p["Xc"], p["W"], "neg_sq_dist", g[tag + "_theta"], np.ones(1), "EI")
    np.testing.assert_allclose(a, g[tag + "_uEI"], rtol=1e-5, atol=1e-12)
    _check_selection(a, order, g[tag + "_sel_uEI"])
    if tag == "cfg1":
        # the reference's own two code paths (batch vs one-candidate-at-a-time pathos helper,
        # uEI_noiseless.py:63-83 vs :85-116) agree only to round-off x cond(Ky)
        np.testing.assert_allclose(g[tag + "_uEI_par"], g[tag + "_uEI"], rtol=1e-6, atol=1e-12)
    a, _, _ = R.batch_uEI(model, p["Xc"], p["W"], "neg_sq_dist", g[tag + "_theta"], np.ones(1), "PI")
    np.testing.assert_allclose(a, g[tag + "_uPI"], rtol=1e-5, atol=1e-12)
    a, order, _ = R.batch_maEI(model, p["Xc"], g[tag + "_theta_lin"], np.ones(1), "EI")
    np.testing.assert_allclose(a, g[tag + "_maEI"], rtol=1e-5, atol=1e-12)
    _check_selection(a, order, g[tag + "_sel_maEI"])


@pytest.mark.parametrize("tag,kind,N,d,m,n,S", [("se", "se", 60, 3, 2, 7, 6), ("rbf", "rbf", 60, 3, 2, 7, 6), ("m52", "matern52", 50, 4, 3, 5, 5)])
def test_gradients(golden, tag, kind, N, d, m, n, S):
    """Gradient rows: kern.gradients_X, posterior mean/variance gradients (gp.py:438-490) and the
    acquisitions' _compute_acq_withGradients, against the reference's own numbers."""
    g = golden("gradients")
    p = R.synthetic_problem(N, d, m, n, S, int(g[tag + "_seed"]), noise=float(g[tag + "_noise"]))
    model = R.MultiOutputGPRef(kind, p["variances"], p["lengthscales"], p["noise"])
    model.updateModel(p["X"], p["Y"])
    Xc = p["Xc"]
    gx = R.kern_gradients_X(kind, g[tag + "_dL"], Xc, p["X"], p["variances"][0], p["lengthscales"][0])
    np.testing.assert_allclose(gx, g[tag + "_gradX"], rtol=1e-10, atol=1e-12)
    dmean, dvar = model.posterior_mean_gradient(Xc), model.posterior_variance_gradient(Xc)
    np.testing.assert_allclose(dmean, g[tag + "_dmean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dvar, g[tag + "_dvar"], rtol=1e-5, atol=1e-8)
    mean, var = model.predict(Xc)
    mu_eval = model.posterior_mean_at_evaluated_points()
    for name, kd in (("maEI", "EI"), ("maPI", "PI")):
        a, da = R.ma_acq_with_gradient(mean, var, dmean, dvar, mu_eval, g[tag + "_support"], g[tag + "_prob"], kd)
        np.testing.assert_allclose(a, g["%s_%s_acq" % (tag, name)], rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(da, g["%s_%s_dacq" % (tag, name)], rtol=1e-5, atol=1e-9)
    mu, sig = model.posterior_mean(Xc), np.sqrt(model.posterior_variance(Xc))
    for uname in ("neg_sq_dist", "neg_sum_exp"):
        sup, pr = (g[tag + "_support"], g[tag + "_prob"]) if uname == "neg_sq_dist" else (np.ones((1, 1)), np.ones(1))
        a, da = R.mc_acq_with_gradient(mu, sig, dmean, dvar, mu_eval, p["W"], uname, sup, pr)
        np.testing.assert_allclose(a, g["%s_uEI_%s_acq" % (tag, uname)], rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(da, g["%s_uEI_%s_dacq" % (tag, uname)], rtol=1e-5, atol=1e-9)
    # finite-difference sanity of the restated mean gradient (independent of the reference)
    h = 1e-6
    for q in range(d):
        Xp, Xm = Xc.copy(), Xc.copy()
        Xp[:, q] += h
        Xm[:, q] -= h
        fd = (model.posterior_mean(Xp) - model.posterior_mean(Xm)) / (2 * h)
        np.testing.assert_allclose(dmean[:, :, q], fd, rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("tag,kind", [("se_ard", "se"), ("se_iso", "se"), ("rbf_ard", "rbf"), ("rbf_iso", "rbf"), ("m52_ard", "matern52"),
                                      ("m32_iso", "matern32")])
def test_hyper_gradients(golden, tag, kind):
    """d log-marginal / d (variance, lengthscale, noise): oracle vs the reference's inference + update_gradients_full,
    plus a finite-difference check of the restatement itself."""
    g = golden("hypergrads")
    X, Y, ls, var, noise = g[tag + "_X"], g[tag + "_Y"], g[tag + "_ls"], float(g[tag + "_var"]), float(g[tag + "_noise"])
    fit = R.GPFit(kind, X, Y, var, ls, noise)
    np.testing.assert_allclose(fit.log_marginal, g[tag + "_lml"], rtol=1e-10)
    dvar, dls, dnoise = fit.lml_gradients()
    np.testing.assert_allclose(dvar, g[tag + "_dvar"], rtol=1e-7)
    np.testing.assert_allclose(dls, g[tag + "_dls"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(dnoise, g[tag + "_dnoise"], rtol=1e-7)
    h = 1e-6
    fd = (R.GPFit(kind, X, Y, var + h, ls, noise).log_marginal - R.GPFit(kind, X, Y, var - h, ls, noise).log_marginal) / (2 * h)
    np.testing.assert_allclose(dvar, fd, rtol=1e-4)
    l2 = ls.copy(); l2[0] += h
    l1 = ls.copy(); l1[0] -= h
    fd = (R.GPFit(kind, X, Y, var, l2, noise).log_marginal - R.GPFit(kind, X, Y, var, l1, noise).log_marginal) / (2 * h)
    want = dls[0] if ls.size > 1 else dls[0]
    np.testing.assert_allclose(want, fd, rtol=1e-4, atol=1e-6)


def test_e2e_config2_full_size(golden):
    """BASELINE configs[1] at full size (m=4 RBF, N=1024, d=6, S=256, C=8192): acquisition values and the
    reference's np.argsort(-acq)[:16] selection (SURVEY 8c G-SEL)."""
    g = golden("e2e")
    N, d, m, C, S = 1024, 6, 4, 8192, 256
    p = R.synthetic_problem(N, d, m, C, S, int(g["cfg2_seed"]))
    model = R.MultiOutputGPRef("rbf", p["variances"], p["lengthscales"], p["noise"])
    model.updateModel(p["X"], p["Y"])
    np.testing.assert_allclose([o.log_marginal for o in model.output], g["cfg2_lml"], rtol=1e-9)
    a, order, _ = R.batch_uEI(model, p["Xc"], p["W"], "neg_sq_dist", g["cfg2_theta"], np.ones(1), "EI")
    np.testing.assert_allclose(a, g["cfg2_uEI"], rtol=1e-5, atol=1e-7 * g["cfg2_uEI"].max())
    _check_selection(g["cfg2_uEI"], order, g["cfg2_sel_uEI"])
    a, order, _ = R.batch_maEI(model, p["Xc"], np.full((1, m), 1.0 / m), np.ones(1), "EI")
    np.testing.assert_allclose(a, g["cfg2_maEI"], rtol=1e-5, atol=1e-7 * g["cfg2_maEI"].max())
    _check_selection(g["cfg2_maEI"], order, g["cfg2_sel_maEI"])


# ---- rank 3: hyper-parameter objective, gradients and HMC chains (golden from the reference's hmc.py / priors.py /
# kernels / inference executed verbatim; paramz's Logexp is the restated piece)
_HYPER = [("se_ard_fixed", "se"), ("rbf_iso_free", "rbf"), ("m52_ard_free", "matern52")]


def _hyper_model(g, tag, kind):
    th, fixed = g[tag + "_theta0"], g[tag + "_fixed"]
    return R.GPHyperRef(kind, g[tag + "_X"], g[tag + "_Y"], th[0], th[1:-1], th[-1], bool(fixed[-1]))


@pytest.mark.parametrize("tag,kind", _HYPER)
def test_hyper_objective_and_gradients_golden(golden, tag, kind):
    g = golden("hyper")
    model = _hyper_model(g, tag, kind)
    np.testing.assert_allclose(model.objective_function(), g[tag + "_obj"], rtol=1e-10)
    np.testing.assert_allclose(model.objective_function_gradients(), g[tag + "_grad"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(model._transform_gradients(model.objective_function_gradients()), g[tag + "_tgrad"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(model.optimizer_array, g[tag + "_x0"], rtol=1e-12)
    # the Logexp restatement is self-consistent and the transformed gradient is the gradient w.r.t. optimizer_array
    x0 = model.optimizer_array.copy()
    np.testing.assert_allclose(R.logexp_f(x0), model.unfixed_param_array, rtol=1e-12)
    tg = model._transform_gradients(model.objective_function_gradients())
    for i in range(x0.size):
        e = np.zeros_like(x0)
        e[i] = 1e-6
        model.optimizer_array = x0 + e
        fp = model.objective_function()
        model.optimizer_array = x0 - e
        fm = model.objective_function()
        np.testing.assert_allclose(tg[i], (fp - fm) / 2e-6, rtol=2e-3, atol=1e-4)


@pytest.mark.parametrize("tag,kind", _HYPER)
def test_hmc_chain_golden(golden, tag, kind):
    """hmc.py:30-69 restated (cpu_ref.hmc_sample) reproduces the reference's seeded chain sample by sample."""
    g = golden("hyper")
    model = _hyper_model(g, tag, kind)
    np.random.seed(int(g[tag + "_seed"]))
    chain = R.hmc_sample(model, int(g[tag + "_num_samples"]), int(g[tag + "_hmc_iters"]), float(g[tag + "_stepsize"]))
    np.testing.assert_allclose(chain, g[tag + "_chain"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(model.param_array, g[tag + "_theta_end"], rtol=1e-6, atol=1e-9)
    assert len(np.unique(chain.round(10), axis=0)) > 3        # the chain moved (accepted proposals)


@pytest.mark.parametrize("tag,accepted", [("N64", 9), ("N256", 0)])
def test_hmc_chain_at_reference_defaults_golden(golden, tag, accepted):
    """The reference's own sampler at ITS defaults (gpmodel.py:31: step_size 0.1, 20 leapfrog steps; exact_feval; default SE-ARD
    kernel), executed under the shim after the optimiser and the 1 % jitter (gpmodel.py:115-118): no exception at either
    size; at N = 64 nine of forty proposals are accepted, at N = 256 NONE of twenty-four -- the reference itself rejects
    every proposal there (the step is too long for the sharper posterior).  The restated sampler reproduces both chains."""
    g = golden("hyper_defaults")
    assert int(g[tag + "_raised_at"]) == -1
    X, Y, th = g[tag + "_X"], g[tag + "_Y"], g[tag + "_theta_start"]
    model = R.GPHyperRef("se", X, Y, th[0], th[1:-1], th[-1], True)
    np.random.seed(int(g[tag + "_seed"]))
    np.random.randn(th.size)                                  # gpmodel.py:116 consumed these before the chain started
    ns = int(g[tag + "_num_samples"])
    chain = R.hmc_sample(model, ns, 20, 1e-1)
    np.testing.assert_allclose(chain, g[tag + "_chain"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(model.param_array, g[tag + "_theta_end"], rtol=1e-5, atol=1e-8)
    moved = np.any(np.diff(np.vstack([th[:-1], chain]), axis=0) != 0, axis=1)       # row i = the state after draw i (hmc.py:45-57)
    assert int(moved.sum()) == accepted


def test_gamma_prior_and_logexp_known_values():
    p = R.GammaPrior.from_EV(2., 4.)                          # gpmodel.py:67: a = 1, b = 0.5
    assert (p.a, p.b) == (1.0, 0.5)
    np.testing.assert_allclose(p.lnpdf(1.3), np.log(0.5) - 0.65, rtol=1e-14)
    np.testing.assert_allclose(p.lnpdf_grad(np.array([0.5, 2.0])), [-0.5, -0.5])
    x = np.array([-40.0, -1.0, 0.0, 3.0, 40.0])
    f = R.logexp_f(x)
    np.testing.assert_allclose(f, [np.exp(-40.0), np.log1p(np.exp(-1.0)), np.log(2.0), np.log1p(np.exp(3.0)), 40.0], rtol=1e-12)
    np.testing.assert_allclose(R.logexp_finv(f)[1:], x[1:], rtol=1e-10)
    np.testing.assert_allclose(R.logexp_gradfactor(f, np.ones(5))[1:4], 1.0 / (1.0 + np.exp(-x[1:4])), rtol=1e-12)


def test_gpmodel_update_restated_flow():
    """gpmodel.py:115-120 on a small problem: optimise, 1 % jitter of every parameter (the fixed noise too), HMC,
    thinning -- shapes, RNG consumption and that the optimiser lowered the objective."""
    rng = np.random.RandomState(3)
    X = rng.uniform(size=(20, 2))
    Y = np.sin(4 * X[:, :1]) + X[:, 1:]
    model = R.GPHyperRef("se", X, Y, 1.0, [1.0, 1.0], 1e-6, True)
    f0 = model.objective_function()
    np.random.seed(10)
    ss = R.gpmodel_update(model, n_samples=4, n_burnin=6, subsample_interval=3, leapfrog_steps=4)
    assert ss.shape == (4, 3) and np.all(ss > 0)
    assert model.param_array[-1] != 1e-6 and abs(model.param_array[-1] / 1e-6 - 1) < 0.1      # raw write jitters the fixed noise
    after = np.random.rand()
    np.random.seed(10)
    np.random.randn(4)
    for _ in range(6 + 4 * 3):
        np.random.multivariate_normal(np.zeros(3), np.eye(3))
        np.random.rand()
    assert after == np.random.rand()                          # exactly randn(P_all) + per sample (mvn(P_free), rand)
    m2 = R.GPHyperRef("se", X, Y, 1.0, [1.0, 1.0], 1e-6, True)
    R.optimize_hyper(m2, 200)
    assert m2.objective_function() < f0 - 1.0


# predict(X, full_cov=True): the oracle's restatement against the reference's PosteriorExact._raw_predict(full_cov=True) + the fork's
# noise / clip / column-0 conventions (oracle/make_golden.py:gen_fullcov), one kernel family per output in the second case
@pytest.mark.parametrize("tag", ["rbf", "mixed"])
def test_predict_full_cov(golden, tag):
    g = golden("fullcov")
    kinds = [str(k) for k in g[tag + "_kinds"]]
    N, d, C, seed = int(g[tag + "_N"]), int(g[tag + "_d"]), int(g[tag + "_C"]), int(g[tag + "_seed"])
    p = R.synthetic_problem(N, d, len(kinds), C, 4, seed)
    ref = R.MultiOutputGPRef(kinds, p["variances"], p["lengthscales"], p["noise"])
    ref.updateModel(p["X"], p["Y"])
    mean, cov = ref.predict(g[tag + "_Xc"], full_cov=True)
    np.testing.assert_allclose(mean, g[tag + "_mean"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(cov, g[tag + "_cov0"], rtol=1e-9, atol=1e-13)
    assert (cov == 1e-10).sum() == (g[tag + "_cov0"] == 1e-10).sum() > 0          # negative covariances are clipped like variances
    np.testing.assert_allclose(cov[:, 0], ref.predict(g[tag + "_Xc"])[1][:, 0], rtol=1e-12, atol=1e-14)
