"""GPU parity, round 4: the resident-team factorization (chol_team.hip: Cholesky + inverse [+ Ky^-1] of 2..8 panels in ONE launch)
against the launched schedule and the oracle; the stream-resident HMC chain (bocf_hmc_streamed) against the lockstep host loop and the
reference's golden chain; the opt-in reproduction of the reference's SE prediction instances.
Run on the MI355X box: python -m pytest tests -m gpu"""
import numpy as np
import pytest

from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import bocf_amd
    bocf_amd._ffi.load()          # fail loudly if the HIP library is missing
    return bocf_amd


def _kern(B, kind, d, var, ls):
    cls = {"rbf": B.kern.RBF, "se": B.kern.SE, "matern52": B.kern.Matern52, "matern32": B.kern.Matern32}[kind]
    ls = np.atleast_1d(ls)
    return cls(d, variance=var, lengthscale=ls, ARD=ls.size > 1)


def _fit(B, kind, p, opts, fits=1):
    d, m = p["X"].shape[1], len(p["Y"])
    model = B.multi_outputGP(m, kernel=[_kern(B, kind, d, p["variances"][j], p["lengthscales"][j]) for j in range(m)], noise_var=list(p["noise"]),
                             fixed_hyps=True)
    model.incremental = False
    for k, v in opts:
        model.set_option(k, v)
    for _ in range(fits):
        model.updateModel(p["X"], p["Y"])
    return model


# ---------------------------------------------------------------------------------------------
# Team schedule (default for 2..24 panels) against the launched single-stream schedule: the factor is the same bit for bit (every tile
# receives the same K = 128 products in the same panel order), the inverse -- column recurrence instead of recursive doubling -- the same
# up to rounding; both against the oracle's LAPACK fit.  Ragged sizes (padding rows), one to five outputs, all four kernel families.
@pytest.mark.parametrize("N,m,kind", [(130, 5, "rbf"), (256, 1, "se"), (300, 3, "matern52"), (700, 2, "matern32"), (1024, 4, "rbf"), (1500, 2, "rbf"),
                                      (2500, 1, "matern52"), (3000, 3, "se")])
def test_team_schedule_equals_launched_schedule(B, N, m, kind):
    d = 5
    p = R.synthetic_problem(N, d, m, 200, 8, 4100 + N, noise=1e-5)
    launched = _fit(B, kind, p, [("team_fit", 0)])
    team = _fit(B, kind, p, [])
    assert launched._context().stat("last_schedule") in (0, 2)
    ctx = team._context()
    assert ctx.stat("last_schedule") == 3 and ctx.stat("sched_timeouts") == 0
    mean0, var0 = launched.predict(p["Xc"])
    mean1, var1 = team.predict(p["Xc"])
    for j in range(m):
        L0, a0 = launched.get_factor(j)
        L1, a1 = team.get_factor(j)
        if N <= 1024:                                       # up to 8 panels: the same K = 128 products in the same order
            np.testing.assert_array_equal(L1, L0)
        else:                                               # beyond: rows are taken several at a time (another summation order)
            np.testing.assert_allclose(L1, L0, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(a1, a0, rtol=1e-7, atol=1e-9 * np.abs(a0).max())
    np.testing.assert_allclose(mean1, mean0, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(var1, var0, rtol=1e-6, atol=1e-11)
    np.testing.assert_allclose(team.log_marginal, launched.log_marginal, rtol=1e-12)
    ref = R.MultiOutputGPRef(kind, p["variances"], p["lengthscales"], p["noise"])
    ref.updateModel(p["X"], p["Y"])
    rm, rv = ref.predict(p["Xc"])
    np.testing.assert_allclose(mean1, rm, rtol=1e-5, atol=1e-6)
    assert np.abs(var1 - rv).max() <= 1e-5 * max(p["variances"]) + 1e-10
    # gradients use R^T (written by the teams) and R (its transpose)
    dm0, dv0 = launched.posterior_mean_gradient(p["Xc"][:7]), launched.posterior_variance_gradient(p["Xc"][:7])
    dm1, dv1 = team.posterior_mean_gradient(p["Xc"][:7]), team.posterior_variance_gradient(p["Xc"][:7])
    np.testing.assert_allclose(dm1, dm0, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(dv1, dv0, rtol=1e-6, atol=1e-11)


# Many factorizations at once (hyper-samples x outputs): more outputs than a launch can give a team of its own -- small teams, and
# above half the compute units several launches.  The probes hook pretends the device is small.
@pytest.mark.parametrize("cus", [0, 24, 8])
def test_team_schedule_small_teams(B, probes, cus):
    N, d, m = 300, 3, 10
    p = R.synthetic_problem(N, d, m, 32, 8, 4242, noise=1e-4)
    launched = _fit(B, "rbf", p, [("team_fit", 0)])
    team = _fit(B, "rbf", p, [("force_cu_count", cus)] if cus else [])
    assert team._context().stat("last_schedule") == 3 and team._context().stat("sched_timeouts") == 0
    for j in range(m):
        np.testing.assert_array_equal(team.get_factor(j)[0], launched.get_factor(j)[0])
    np.testing.assert_allclose(team.predict(p["Xc"])[1], launched.predict(p["Xc"])[1], rtol=1e-7, atol=1e-12)


# The jitter ladder through the team schedule (linalg.py:52-71): the device reports the first bad pivot per output, the host climbs the
# ladder and re-launches; and a dependency time-out (forced by the probes hook) sends the attempt back to the launched schedule.
def test_team_schedule_jitter_ladder_and_fallback(B, probes):
    rng = np.random.RandomState(7)
    N, d = 200, 2
    X = rng.uniform(size=(N, d))
    Y = np.sin(3 * X.sum(1))[:, None]
    var, ls, noise = 1.0, np.array([0.9]), 1e-6
    K = R.kern_K("se", X, None, var, ls)
    lam = np.linalg.eigvalsh(K)
    shift = noise + 1e-8 + max(lam.min(), 0.0) + 3e-4
    Ky = K + (noise + 1e-8 - shift) * np.eye(N)
    Lref, jref = R.jitchol(Ky)
    assert jref > 0
    model = B.multi_outputGP(1, kernel=[_kern(B, "se", d, var, ls)], noise_var=[noise], fixed_hyps=True)
    model.incremental = False
    model.set_option("test_diag_shift_1e12", int(round(shift * 1e12)))
    model.updateModel(X, [Y])
    assert model._context().stat("last_schedule") == 3
    assert model.jitter[0] == pytest.approx(jref, rel=1e-12)
    L, _ = model.get_factor(0)
    np.testing.assert_allclose(L.dot(L.T), Ky + jref * np.eye(N), rtol=0, atol=1e-10)
    # time-out (no ladder in the way: each rung is an attempt of its own and only the one that timed out is redone): redone on the launched
    # schedule, counted, same factor; the next fit is the teams' again
    model.set_option("test_diag_shift_1e12", 0)
    model.updateModel(X, [Y])
    ctx = model._context()
    assert ctx.stat("last_schedule") == 3 and model.jitter[0] == 0.0
    L = model.get_factor(0)[0]
    model.set_option("force_sched_timeout", 1)
    model.updateModel(X, [Y])
    assert ctx.stat("sched_timeouts") == 1 and ctx.stat("last_schedule") == 0
    np.testing.assert_array_equal(model.get_factor(0)[0], L)
    model.updateModel(X, [Y])
    assert ctx.stat("sched_timeouts") == 1 and ctx.stat("last_schedule") == 3


# Inference mode: the teams also accumulate Ky^-1 = R R^T (pdinv's dpotri, linalg.py:206) underneath the factorization; log-marginal and
# hyper-gradients against the launched path (factorization, then a triangular GEMM) and the oracle's closed forms.
@pytest.mark.parametrize("N,kind", [(200, "rbf"), (640, "matern52")])
def test_team_inference_equals_launched_inference(B, N, kind):
    d, m = 4, 3
    p = R.synthetic_problem(N, d, m, 8, 8, 4300 + N, noise=1e-3)
    res = []
    for team in (0, -1):
        model = B.multi_outputGP(m, kernel=[_kern(B, kind, d, p["variances"][j], p["lengthscales"][j]) for j in range(m)], noise_var=list(p["noise"]),
                                 fixed_hyps=False, n_samples=2)
        model.set_option("team_fit", team)
        model._X, model._Y = p["X"], p["Y"]
        model._create_sampler_state()
        params = [(p["variances"][j], np.asarray(p["lengthscales"][j], dtype=float), p["noise"][j]) for j in range(m)]
        res.append(model._infer(params))
        assert model._context().stat("last_schedule") == (3 if team else 0)
    for a, b in zip(res[1], res[0]):
        np.testing.assert_allclose(a, b, rtol=1e-8, atol=1e-8 * np.abs(b).max())
    for j in range(m):
        fit = R.GPFit(kind, p["X"], p["Y"][j], p["variances"][j], p["lengthscales"][j], p["noise"][j])
        np.testing.assert_allclose(res[1][0][j], fit.log_marginal, rtol=1e-9)
        dv, dl, dn = fit.lml_gradients()
        np.testing.assert_allclose(res[1][1][j], dv, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(res[1][2][j], dl, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(res[1][3][j], dn, rtol=1e-6, atol=1e-4 * abs(dn))


# ---------------------------------------------------------------------------------------------
# The STREAM-RESIDENT HMC chain (bocf_hmc_streamed: N > 128) against the lockstep host loop (one bocf_infer per leapfrog step, the O(P)
# arithmetic in NumPy): same chains, same accept decisions, same final parameters, one batched inference per step plus the two at the start.
@pytest.mark.parametrize("N,ard,free_noise", [(160, True, False), (300, False, True)])
def test_streamed_hmc_equals_lockstep(B, N, ard, free_noise):
    from bocf_amd import hyper as H
    rng = np.random.RandomState(15)
    d, m, ns, iters, step = 3, 3, 9, 5, 0.02
    X = rng.uniform(size=(N, d))
    Ys = [np.sin(3 * X.dot(rng.normal(size=d)))[:, None] + 0.05 * rng.normal(size=(N, 1)) for _ in range(m)]
    res = []
    for path in ("lockstep", "streamed"):
        model = B.multi_outputGP(m, fixed_hyps=False, n_samples=2, ARD=[ard] * m, exact_feval=[not free_noise] * m)
        model._X, model._Y = X, Ys
        model._create_sampler_state()
        outs = model._sampler_outputs
        draws = H.LockstepSampler.draw(outs, ns, rng=np.random.RandomState(19))
        sampler = H.LockstepSampler(outs, model._infer, d, device_hmc=model._device_hmc if path == "streamed" else None)
        chains = sampler.hmc([dr[1] for dr in draws], [dr[2] for dr in draws], hmc_iters=iters, stepsize=step)
        res.append((chains, sampler.accepted.copy(), [o.param_array.copy() for o in outs], sampler.n_inferences, sampler.diverged.copy()))
    for j in range(m):
        np.testing.assert_allclose(res[1][0][j], res[0][0][j], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(res[1][2][j], res[0][2][j], rtol=1e-7, atol=1e-10)
    np.testing.assert_array_equal(res[1][1], res[0][1])
    np.testing.assert_array_equal(res[1][4], res[0][4])
    assert res[0][1].sum() > 0
    assert res[1][3] == 1 + ns * iters                      # (the staging fit at the start is not one of the chain's evaluations)


# A draw that needs jitchol's ladder is the host's: the streamed chain hands it over (draws_done < num_samples), the host runs that draw
# with bocf_infer per step (ladder included), the device takes the next one.  Forced for EVERY draw with the diagonal-shift hook: the
# result must still be the lockstep chain.
def test_streamed_hmc_hands_ladder_draws_to_the_host(B, probes):
    from bocf_amd import hyper as H
    rng = np.random.RandomState(21)
    N, d, m, ns, iters = 150, 2, 2, 4, 3
    X = rng.uniform(size=(N, d))
    Ys = [np.sin(3 * X.sum(1))[:, None] + 0.01 * rng.normal(size=(N, 1)) for _ in range(m)]
    res = []
    for path in ("lockstep", "streamed"):
        model = B.multi_outputGP(m, fixed_hyps=False, n_samples=2, exact_feval=[True] * m)
        model._X, model._Y = X, Ys
        model._create_sampler_state()
        outs = model._sampler_outputs
        draws = H.LockstepSampler.draw(outs, ns, rng=np.random.RandomState(23))
        sampler = H.LockstepSampler(outs, model._infer, d, device_hmc=model._device_hmc if path == "streamed" else None)
        sampler.evaluate()
        model.set_option("test_diag_shift_1e12", int(round((1e-6 + 1e-8 + 2e-5) * 1e12)))     # Ky loses 2e-5 on its diagonal: rungs 1e-6, 1e-5 fail for smooth kernels
        sampler._key = None
        chains = sampler.hmc([dr[1] for dr in draws], [dr[2] for dr in draws], hmc_iters=iters, stepsize=0.01, on_failure="reject")
        model.set_option("test_diag_shift_1e12", 0)
        res.append((chains, sampler.accepted.copy(), sampler.diverged.copy()))
    for j in range(m):
        np.testing.assert_allclose(res[1][0][j], res[0][0][j], rtol=1e-7, atol=1e-10)
    np.testing.assert_array_equal(res[1][1], res[0][1])
    np.testing.assert_array_equal(res[1][2], res[0][2])


# ---------------------------------------------------------------------------------------------
# The reference's model-instance quirk, opt-in: with fixed_hyps=False and a USER kernel the reference learns the
# hyper-parameters with that kernel but predicts with SE instances carrying the sampled parameters (gpmodel.py:57-61 resets self.kernel,
# :80-84 builds SE, :121-126 writes the HMC samples in).  reference_instance_kernels=True reproduces that: the posterior equals an exact SE
# GP (the reference's se.py, pinned by tests/golden/kernels.npz + fit_predict.npz through the oracle) at the sampled hyper-parameters;
# the default keeps the learned family.
def test_reference_instance_kernels_opt_in(B):
    rng = np.random.RandomState(31)
    N, d, m = 40, 2, 2
    X = rng.uniform(size=(N, d))
    Ys = [np.sin(3 * X.dot(rng.normal(size=d)))[:, None] for _ in range(m)]
    Xc = rng.uniform(size=(50, d))
    out = {}
    for flag in (False, True):
        np.random.seed(77)
        model = B.multi_outputGP(m, kernel=[B.kern.Matern52(d, variance=1.0, lengthscale=np.full(d, 0.7), ARD=True) for _ in range(m)],
                                 fixed_hyps=False, n_samples=2, exact_feval=[True] * m, reference_instance_kernels=flag)
        model.n_burnin, model.subsample_interval, model.leapfrog_steps, model.max_iters = 2, 1, 3, 5
        model.updateModel(X, Ys)
        out[flag] = (model, [s.copy() for s in model.hmc_samples])
    for j in range(m):                                      # the LEARNING is the user's kernel either way: same samples
        np.testing.assert_array_equal(out[True][1][j], out[False][1][j])
    for flag, kind in ((False, "matern52"), (True, "se")):
        model, samples = out[flag]
        for h in range(2):
            model.set_hyperparameters(h)
            mean, var = model.predict(Xc)
            for j in range(m):
                th = samples[j][h]
                fit = R.GPFit(kind, X, Ys[j], th[0], th[1:1 + d], 1e-6)
                rm, rv = fit.predict(Xc)
                np.testing.assert_allclose(mean[j], rm[:, 0], rtol=1e-6, atol=1e-7)
                np.testing.assert_allclose(var[j], np.clip(rv[:, 0], 1e-10, np.inf), rtol=1e-5, atol=1e-9)
    assert np.abs(out[True][0].predict(Xc)[0] - out[False][0].predict(Xc)[0]).max() > 1e-6      # the two really differ


# Beyond 8 panels the teams take rows several at a time, in the background, in whatever order the hand-offs arrive -- but WHICH rows share a
# product is fixed per unit, so the factor does not depend on timing or on the size of the teams: bit-identical from run to run, and between
# teams of 64 and (device pretended small) 8 workgroups.
def test_team_schedule_is_deterministic_beyond_eight_panels(B, probes):
    N, d, m = 1500, 4, 2
    p = R.synthetic_problem(N, d, m, 16, 8, 4747, noise=1e-5)
    models = [_fit(B, "rbf", p, [], fits=2), _fit(B, "rbf", p, []), _fit(B, "rbf", p, [("force_cu_count", 16)])]
    for mod in models:
        assert mod._context().stat("last_schedule") == 3 and mod._context().stat("sched_timeouts") == 0
    for j in range(m):
        L0, a0 = models[0].get_factor(j)
        for mod in models[1:]:
            L1, a1 = mod.get_factor(j)
            np.testing.assert_array_equal(L1, L0)
            np.testing.assert_array_equal(a1, a0)
    v0 = models[0].predict(p["Xc"])[1]
    for mod in models[1:]:
        np.testing.assert_array_equal(mod.predict(p["Xc"])[1], v0)


# ---------------------------------------------------------------------------------------------
# The streamed critical chain (a workgroup of the team forms U[p][p+1] and the last row of A[p+1][p+1] sixteen rows at a time underneath
# potrf(p)) against the same teams without it: up to eight panels the same K = 128 sums in the same order -- bit for bit, whatever the size of
# the teams; beyond, the diagonal units' rows are grouped differently (their last row is the streamer's): equal to rounding.
@pytest.mark.parametrize("N,m,cus", [(300, 2, 0), (1024, 4, 0), (900, 3, 12), (1024, 2, 8), (2000, 2, 0), (2600, 1, 24)])
def test_streamed_critical_chain_equals_unit_products(B, probes, N, m, cus):
    p = R.synthetic_problem(N, 4, m, 32, 8, 5200 + N, noise=1e-5)
    opts = [("force_cu_count", cus)] if cus else []
    plain = _fit(B, "matern52", p, opts + [("team_stream", 0)])
    streamed = _fit(B, "matern52", p, opts)
    for mod in (plain, streamed):
        assert mod._context().stat("last_schedule") == 3 and mod._context().stat("sched_timeouts") == 0
    for j in range(m):
        L0, a0 = plain.get_factor(j)
        L1, a1 = streamed.get_factor(j)
        if N <= 1024:
            np.testing.assert_array_equal(L1, L0)
            np.testing.assert_array_equal(a1, a0)
        else:
            np.testing.assert_allclose(L1, L0, rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(a1, a0, rtol=1e-7, atol=1e-9 * np.abs(a0).max())
    v0, v1 = plain.predict(p["Xc"])[1], streamed.predict(p["Xc"])[1]
    if N <= 1024:
        np.testing.assert_array_equal(v1, v0)
    else:
        np.testing.assert_allclose(v1, v0, rtol=1e-6, atol=1e-11)


# ---------------------------------------------------------------------------------------------
# Small batches travel through pinned staging buffers (asynchronous copies, one synchronisation per call), large ones through the plain
# copies: the same values on both sides of the 256-KiB boundary, for every entry point that returns per-candidate arrays, and when the same
# buffer is reused call after call (and by two set_candidates in a row without anything reading in between).
def test_pinned_staging_boundaries(B):
    N, d, m = 200, 4, 2
    p = R.synthetic_problem(N, d, m, 9000, 16, 6161, noise=1e-4)
    model = _fit(B, "rbf", p, [])
    theta = np.array([[0.3, -0.2]])
    acq = B.maEI(model, None, utility=B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), linear=True))
    Xall = p["Xc"]                                          # 9000 x 4 doubles = 281 KiB: the plain copies (results do not depend on the batch)
    mean_all, var_all = model.predict(Xall)
    a_all = acq.acquisition_function(Xall)
    f_all, g_all = acq.acquisition_function_withGradients(Xall[:4000])      # (4000 x 5 doubles = 156 KiB staged; 8000 x 5 would not be)
    dm_all = model.posterior_mean_gradient(Xall[:8500])    # 2 x 8500 x 4 doubles: plain
    # candidates: 8192 x 4 doubles = 256 KiB exactly (staged), one more row (plain); predict's outputs 2 x 2 x C doubles: staged up to 8192 too
    for C in (1, 7, 8192, 8193, 1638, 3):
        X = Xall[:C]
        model._set_candidates(Xall[C:2 * C] if 2 * C <= 9000 else Xall[:1])     # (overwritten by the next call before anything reads it)
        mean, var = model.predict(X)
        np.testing.assert_allclose(mean, mean_all[:, :C], rtol=1e-10, atol=1e-12)       # (small batches run the wave-level kernels: other sums)
        np.testing.assert_allclose(var, var_all[:, :C], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(acq.acquisition_function(X), a_all[:C], rtol=1e-8, atol=1e-13)
        if C <= 4000:
            f, g = acq.acquisition_function_withGradients(X)
            np.testing.assert_allclose(f, f_all[:C], rtol=1e-8, atol=1e-13)
            np.testing.assert_allclose(g, g_all[:C], rtol=1e-7, atol=1e-11)
        np.testing.assert_allclose(model.posterior_mean_gradient(X), dm_all[:, :C], rtol=1e-9, atol=1e-12)


# ---------------------------------------------------------------------------------------------
# Option predict_i8: the variance contraction in exact int8 digit products (gemm_i8.hip) against the fp64 contraction -- the same posterior
# mean bit for bit (it does not go through the contraction), variances within 1e-9 sigma_f^2 (SURVEY 8c's tolerance is 1e-8 sigma_f^2 + 1e-10; measured 2e-12 ... 1.1e-11), acquisition values and the top-16 with them; ragged N (padding rows), ragged candidate counts, several outputs and kernel families,
# chunked batches, both row-tile groupings; gradients and small batches keep the fp64 path.
@pytest.mark.parametrize("N,m,C,kind", [(100, 2, 200, "rbf"), (300, 1, 700, "matern52"), (130, 2, 300, "rbf"), (700, 3, 1000, "matern52"), (1024, 4, 8192, "rbf"), (2500, 1, 5000, "matern32"), (1900, 2, 40000, "se")])
def test_int8_variance_contraction_against_fp64(B, N, m, C, kind):
    d = 5
    p = R.synthetic_problem(N, d, m, C, 16, 7300 + N, noise=1e-5)
    model = _fit(B, kind, p, [])
    theta = np.array([[0.2 * (j + 1) for j in range(m)]])
    U = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), device="neg_sq_dist")
    acq = B.uEI_noiseless(model, None, utility=U)
    acq.W_samples = p["W"]
    mean0, var0 = model.predict(p["Xc"])
    a0 = acq.acquisition_function(p["Xc"])
    g0 = model.posterior_variance_gradient(p["Xc"][:40])
    s0 = model.predict(p["Xc"][:9])
    out = []
    for grp, chunk in ((0, None), (1, None), (3, 4096)):
        model.set_option("predict_i8", 1)
        model.set_option("i8_group", grp)
        if chunk:
            model.set_option("chunk", chunk)
        mean1, var1 = model.predict(p["Xc"])
        np.testing.assert_array_equal(mean1, mean0)
        vmax = max(p["variances"])
        assert np.abs(var1 - var0).max() <= 1e-9 * vmax, np.abs(var1 - var0).max()
        a1 = acq.acquisition_function(p["Xc"])
        np.testing.assert_allclose(a1, a0, rtol=1e-4, atol=1e-7 * np.abs(a0).max() + 1e-12)
        # what does not go through the big contraction is untouched: gradients (they need V itself) and batches up to 16 points
        np.testing.assert_array_equal(model.posterior_variance_gradient(p["Xc"][:40]), g0)
        s1 = model.predict(p["Xc"][:9])
        np.testing.assert_array_equal(s1[1], s0[1])
        out.append(var1)
    np.testing.assert_array_equal(out[1], out[0])           # the grouping of the row tiles is a matter of speed only
    np.testing.assert_array_equal(out[2], out[0])           # ... and so is the chunking
    model.set_option("predict_i8", 0)
    np.testing.assert_array_equal(model.predict(p["Xc"])[1], var0)


# ... with kernel variances away from 1 (the scale of K*'s digits is 2^ceil(log2 variance)), per-output kernel families, and a refit with other
# hyper-parameters (R's digits and the variance exponents belong to the factorization)
def test_int8_variance_contraction_scales_and_hyper_samples(B):
    N, d, m, C = 600, 4, 3, 2000
    p = R.synthetic_problem(N, d, m, C, 16, 7411, noise=1e-4)
    var = [0.3, 2.5, 17.0]
    kinds = ["rbf", "matern52", "matern32"]
    model = B.multi_outputGP(m, kernel=[_kern(B, kinds[j], d, var[j], p["lengthscales"][j]) for j in range(m)], noise_var=list(p["noise"]), fixed_hyps=True)
    model.incremental = False
    model.updateModel(p["X"], p["Y"])
    mean0, var0 = model.predict(p["Xc"])
    model.set_option("predict_i8", 1)
    mean1, var1 = model.predict(p["Xc"])
    np.testing.assert_array_equal(mean1, mean0)
    for j in range(m):
        assert np.abs(var1[j] - var0[j]).max() <= 1e-9 * var[j], (j, np.abs(var1[j] - var0[j]).max())
    # a refit with other hyper-parameters must rebuild R's digits
    model2_var = [1.9, 0.6, 4.0]
    model.kernel = [_kern(B, kinds[j], d, model2_var[j], 0.7 * np.atleast_1d(p["lengthscales"][j])) for j in range(m)]
    model.updateModel(p["X"], p["Y"])
    _, v_i8 = model.predict(p["Xc"])
    model.set_option("predict_i8", 0)
    _, v_64 = model.predict(p["Xc"])
    for j in range(m):
        assert np.abs(v_i8[j] - v_64[j]).max() <= 1e-9 * model2_var[j]
        assert np.abs(v_64[j] - var0[j]).max() > 1e-6            # (it IS another model)
