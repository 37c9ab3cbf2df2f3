"""GPU parity, round 3: parity at BASELINE configs[2] (N = 4096, cond(Ky) ~ 4e9) judged against an
EXTENDED-PRECISION truth (oracle/truth_ld.c: x87 long double end to end) instead of against the fp64 oracle's own rounding,
for the default fit schedule AND for legal re-orderings of it (panel aggregation G = 1, 2, 3, 4); acquisition parity on
workloads whose acquisition is dense (most candidates have a non-zero EI / PI) at configs[1] against the REFERENCE's numbers
and at configs[2] against the oracle; the fp32 contraction's top-16 set at configs[4].
Run on the MI355X box: python -m pytest tests -m gpu"""
import numpy as np
import pytest

from oracle import cpu_ref as R
from oracle import truth as T

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


def _model(B, kind, X, Ys, variances, lengthscales, noises, options=(), **kw):
    d, m = X.shape[1], len(Ys)
    model = B.multi_outputGP(m, kernel=[_kern(B, kind, d, variances[j], lengthscales[j]) for j in range(m)], noise_var=list(noises),
                             fixed_hyps=True, **kw)
    for name, value in options:
        model.set_option(name, value)
    model.updateModel(X, Ys)
    return model


def _utility(B, theta, device):
    return B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), device=device)


@pytest.fixture(scope="module")
def cfg3(B):
    """BASELINE configs[2] (m=4 RBF ARD, N=4096, d=8, S=1024): the synthetic problem of SURVEY 8(d), a 64-candidate slice
    (48 random candidates + the device's 16 best of the 65 536), the oracle's fit and the long-double truth on that slice."""
    N, d, m, C, S = 4096, 8, 4, 65536, 1024
    p = R.synthetic_problem(N, d, m, C, S, 1237)
    theta = np.array([[0.2 * (j + 1) for j in range(m)]])
    model = _model(B, "rbf", p["X"], p["Y"], p["variances"], p["lengthscales"], p["noise"])
    acq = B.uEI_noiseless(model, None, utility=_utility(B, theta, "neg_sq_dist"))
    acq.W_samples = p["W"]
    a = acq._compute_acq(p["Xc"])
    idx = np.concatenate([np.arange(48), np.argsort(-a[:, 0], kind="stable")[:16]])
    Xs = p["Xc"][idx]
    ref = R.MultiOutputGPRef("rbf", p["variances"], p["lengthscales"], p["noise"])
    ref.updateModel(p["X"], p["Y"])
    tru = T.model_truth("rbf", p["X"], p["Y"], p["variances"], p["lengthscales"], p["noise"], Xs)
    return dict(p=p, theta=theta, idx=idx, Xs=Xs, ref=ref, tru=tru, m=m)


def _acq_from_posterior(mu, var, mu_train, W, theta):
    return R.mc_acq(mu, np.sqrt(var), mu_train, W, "neg_sq_dist", theta, np.ones(1), "EI")[0][:, 0]


# ---------------------------------------------------------------------------------------------
# The gate: err_device <= max(1e-5 * scale, 4 * err_oracle), both errors measured against the long-double truth
# (posterior.py:299-320, uEI_noiseless.py:63-83).  scale: |mean| -> max |truth mean|; variance -> sigma_f^2 = 1 (the quantity the
# cancellation k** - ||v||^2 is carried at); acquisition -> its maximum over the slice.  Every legal re-ordering of the
# factorization (panels per trailing update G = 1 ... 4: other summation orders of the same sums) must pass the SAME gate:
# that is what makes the choice of schedule a matter of speed only.
@pytest.mark.parametrize("G", [0, 1, 2, 3, 4, "i8"])
def test_config3_against_extended_precision_truth(B, cfg3, G):
    p, Xs, ref, tru, theta = cfg3["p"], cfg3["Xs"], cfg3["ref"], cfg3["tru"], cfg3["theta"]
    i8 = G == "i8"                                  # the variance contraction in exact int8 digit products (option predict_i8), default schedule
    G = 0 if i8 else G
    model = _model(B, "rbf", p["X"], p["Y"], p["variances"], p["lengthscales"], p["noise"],
                   options=[("aggregate", G), ("lookahead", 0 if G else -1)] + ([("predict_i8", 1)] if i8 else []))
    mean, var = model.predict(Xs)
    mu_tr = model.posterior_mean_at_evaluated_points()
    rm, rv = ref.predict(Xs)
    r_tr = ref.posterior_mean_at_evaluated_points()
    rows = []
    ok_all = True
    for name, dev, orc, tr, scale in [("mean", mean, rm, tru["mean"], np.abs(tru["mean"]).max()),
                                      ("variance", var, rv, tru["var"], 1.0),
                                      ("mean at train", mu_tr, r_tr, tru["mu_train"], np.abs(tru["mu_train"]).max())]:
        ok, e_dev, e_orc, bound = T.gate(dev, orc, tr, scale)
        rows.append("%-14s device %.3e   oracle %.3e   bound %.3e" % (name, e_dev, e_orc, bound))
        ok_all &= ok
    rel_dev = (np.abs(var - tru["var"]) / tru["var"]).max()
    rel_orc = (np.abs(rv - tru["var"]) / tru["var"]).max()
    rows.append("variance RELATIVE (values %.2e..%.2e): device %.3e   oracle %.3e" % (tru["var"].min(), tru["var"].max(), rel_dev, rel_orc))
    # acquisition: the device's own kernel chain against the oracle's Monte-Carlo on the TRUE posterior
    acq = B.uEI_noiseless(model, None, utility=_utility(B, theta, "neg_sq_dist"))
    acq.W_samples = p["W"]
    a_dev = acq._compute_acq(Xs)[:, 0]
    a_orc = _acq_from_posterior(rm, rv, r_tr, p["W"], theta)
    a_tru = _acq_from_posterior(tru["mean"], tru["var"], tru["mu_train"], p["W"], theta)
    ok, e_dev, e_orc, bound = T.gate(a_dev, a_orc, a_tru, a_tru.max())
    rows.append("%-14s device %.3e   oracle %.3e   bound %.3e   (max %.3e, %d non-zero)" % ("uEI", e_dev, e_orc, bound, a_tru.max(), int((a_tru > 0).sum())))
    ok_all &= ok
    print("config 3, aggregate = %d, max abs error against the long-double truth:\n  " % G + "\n  ".join(rows))
    assert ok_all
    # the variance RELATIVE to itself, at variances of 1e-6 sigma_f^2: fp64 contraction 1e-8; the int8 contraction's six radix-254 digits carry
    # 47.9 bits at the scale of R's columns: 3.4e-6 (its first form -- radix 128, 41 bits -- gave 1.3e-4 and failed this line)
    assert rel_dev <= max(1e-5, 4 * rel_orc)
    assert np.argmax(a_dev) == np.argmax(a_tru)
    np.testing.assert_allclose(model.log_marginal, tru["lml"], rtol=1e-9)


# ---------------------------------------------------------------------------------------------
# Dense acquisition at configs[2] shape: targets scaled by 1e-3 put the spread of the posterior mean at the scale of the
# posterior standard deviation, so most of a random 512-candidate slice has EI > 1e-3 max (on the unscaled targets FOUR of the 512
# best have a non-zero value).  Device vs oracle at rtol 1e-5 on every candidate above 1e-3 max, the top-16 SET identical, for
# uEI_noiseless / uPI (uEI_noiseless.py:63-83, uPI.py:66-86) with two utilities and maEI / maPI (maEI.py:81-98, maPI.py:78-94);
# and the same values against the truth through the gate.
def test_config3_shape_dense_acquisition(B):
    N, d, m, C, S, amp = 4096, 8, 4, 512, 1024, 1e-3
    p = R.synthetic_problem(N, d, m, C, S, 1237)
    Ys = [amp * y for y in p["Y"]]
    model = _model(B, "rbf", p["X"], Ys, p["variances"], p["lengthscales"], p["noise"])
    ref = R.MultiOutputGPRef("rbf", p["variances"], p["lengthscales"], p["noise"])
    ref.updateModel(p["X"], Ys)
    tru = T.model_truth("rbf", p["X"], Ys, p["variances"], p["lengthscales"], p["noise"], p["Xc"])
    theta = np.array([[0.2 * (j + 1) for j in range(m)]])
    th_lin = np.full((1, m), 1.0 / m)
    rm, rv = ref.predict(p["Xc"])
    r_tr = ref.posterior_mean_at_evaluated_points()
    lines = []
    for name, cls, util, kind in [("uEI neg_sq_dist", B.uEI_noiseless, "neg_sq_dist", "EI"), ("uEI neg_sum_exp", B.uEI_noiseless, "neg_sum_exp", "EI"),
                                  ("uPI neg_sq_dist", B.uPI, "neg_sq_dist", "PI"), ("maEI", B.maEI, None, "EI"), ("maPI", B.maPI, None, "PI")]:
        if util is None:
            U = B.Utility(parameter_dist=B.ParameterDistribution(support=th_lin, prob_dist=np.ones(1)), linear=True)
            acq = cls(model, None, utility=U)
            r = R.ma_acq(rm, rv, r_tr, th_lin, np.ones(1), kind)[0]
            t = R.ma_acq(tru["mean"], tru["var"], tru["mu_train"], th_lin, np.ones(1), kind)[0]
        else:
            acq = cls(model, None, utility=_utility(B, theta, util))
            acq.W_samples = p["W"]
            r = R.mc_acq(rm, np.sqrt(rv), r_tr, p["W"], util, theta, np.ones(1), kind)[0]
            t = R.mc_acq(tru["mean"], np.sqrt(tru["var"]), tru["mu_train"], p["W"], util, theta, np.ones(1), kind)[0]
        a = acq._compute_acq(p["Xc"])
        big = r[:, 0] > 1e-3 * r.max()
        dense = float(big.mean())
        rel = (np.abs(a[big, 0] - r[big, 0]) / r[big, 0]).max()
        ok, e_dev, e_orc, bound = T.gate(a, r, t, t.max())
        lines.append("%-16s max %.3e  dense %.2f  max rel err vs oracle (above 1e-3 max) %.2e | vs truth: device %.2e oracle %.2e bound %.2e" %
                     (name, r.max(), dense, rel, e_dev, e_orc, bound))
        assert dense >= 0.25, (name, dense)
        if kind == "PI" and util is not None:
            # Monte-Carlo PI is a count / S: a sample within rounding of the threshold may flip on either side
            assert np.abs(a - r).max() <= 1.0 / S + 1e-12, name
        else:
            assert rel < 1e-5, (name, rel)
        assert ok, (name, e_dev, e_orc, bound)
        top = acq.select_anchors(16)
        assert set(top.tolist()) == set(np.argsort(-r[:, 0], kind="stable")[:16].tolist()), name
    print("config-3 shape, dense workload (targets x 1e-3):\n  " + "\n  ".join(lines))


# ---------------------------------------------------------------------------------------------
# The same at configs[1] FULL size against the REFERENCE's own classes (tests/golden/dense.npz, oracle/make_golden.py:gen_dense):
# uEI_noiseless / uPI / maEI / maPI executed verbatim on the reference's inference + posterior, 8192 candidates.
def test_config2_dense_golden(B, golden):
    g = golden("dense")
    N, d, m, C, S = 1024, 6, 4, 8192, 256
    p = R.synthetic_problem(N, d, m, C, S, int(g["cfg2d_seed"]))
    Ys = [float(g["cfg2d_amp"]) * y for y in p["Y"]]
    model = _model(B, "rbf", p["X"], Ys, p["variances"], p["lengthscales"], p["noise"])
    lines = []
    for name, cls, mc in [("uEI", B.uEI_noiseless, True), ("uPI", B.uPI, True), ("maEI", B.maEI, False), ("maPI", B.maPI, False)]:
        if mc:
            acq = cls(model, None, utility=_utility(B, g["cfg2d_theta"], "neg_sq_dist"))
            acq.W_samples = p["W"]
        else:
            acq = cls(model, None, utility=B.Utility(parameter_dist=B.ParameterDistribution(support=g["cfg2d_theta_lin"], prob_dist=np.ones(1)), linear=True))
        a = acq._compute_acq(p["Xc"])
        r = g["cfg2d_" + name]
        big = r[:, 0] > 1e-3 * r.max()
        dense = float(big.mean())
        rel = (np.abs(a[big, 0] - r[big, 0]) / r[big, 0]).max()
        lines.append("%-5s max %.3e  dense %.2f  max rel err above 1e-3 max: %.2e   max abs err: %.2e" % (name, r.max(), dense, rel, np.abs(a - r).max()))
        assert dense >= 0.25, (name, dense)
        if name == "uPI":
            assert np.abs(a - r).max() <= 1.0 / S + 1e-12
        else:
            assert rel < 1e-5, (name, rel)
            np.testing.assert_allclose(a, r, rtol=1e-5, atol=1e-8 * r.max())
        top = acq.select_anchors(16)
        ref_sel = g["cfg2d_sel_" + name]
        # the 16 best of the reference, allowing a swap only between values that agree to the gate
        assert set(top.tolist()) == set(ref_sel.tolist()) or np.allclose(np.sort(a[top, 0]), np.sort(r[ref_sel, 0]), rtol=1e-5), name
    print("config 2 dense workload vs the REFERENCE's values:\n  " + "\n  ".join(lines))


# ---------------------------------------------------------------------------------------------
# BASELINE configs[4] with the fp32 contraction: the top-16 SET of a slice against the fp64 oracle (VERDICT r2 "weak": the arg-max
# was asserted for the fp64 run only).  Slice = 192 random candidates + the 64 best of the batch; the acquisition on the dense
# workload (targets x 5e-2: sigma here is 3e-2 ... 2e-1) so that the ranking is decided by values the fp32 rounding could move.
def test_config5_fp32_top16_set(B):
    N, d, m, C, S, amp = 8192, 12, 8, 8192, 4096, 5e-2
    p = R.synthetic_problem(N, d, m, C, S, 1239, noise=1e-4)
    Ys = [amp * y for y in p["Y"]]
    model = _model(B, "matern52", p["X"], Ys, p["variances"], p["lengthscales"], p["noise"])
    theta = np.array([[0.1 * (j + 1) for j in range(m)]])
    acq = B.uEI_noiseless(model, None, utility=_utility(B, theta, "neg_sq_dist"))
    acq.W_samples = p["W"]
    a64 = acq._compute_acq(p["Xc"])
    model.set_option("predict_f32", 1)
    a32 = acq._compute_acq(p["Xc"])
    idx = np.concatenate([np.arange(192), np.argsort(-a64[:, 0], kind="stable")[:64]])
    idx = np.unique(idx)
    ref = R.MultiOutputGPRef("matern52", p["variances"], p["lengthscales"], p["noise"])
    ref.updateModel(p["X"], Ys)
    r, _, _ = R.batch_uEI(ref, p["Xc"][idx], p["W"], "neg_sq_dist", theta, np.ones(1), "EI")
    dense = float((r[:, 0] > 1e-3 * r.max()).mean())
    rel64 = np.abs(a64[idx] - r).max() / r.max()
    rel32 = np.abs(a32[idx] - r).max() / r.max()
    print("config 5 dense workload: %d-candidate slice, dense %.2f, max |d acq| / max: fp64 %.2e  fp32 contraction %.2e" % (len(idx), dense, rel64, rel32))
    assert dense >= 0.25
    order = np.argsort(-r[:, 0], kind="stable")
    ref16, cut = idx[order[:16]], r[order[15], 0]
    rpos = {int(i): float(v) for i, v in zip(idx, r[:, 0])}
    for name, a, tol in (("fp64", a64, 1e-5 * r.max()), ("fp32", a32, 5e-3 * r.max())):
        dev16 = idx[np.argsort(-a[idx, 0], kind="stable")[:16]]
        # the SET of the 16 best: a member may differ from the oracle's only where the oracle's own value is within the
        # arithmetic's documented tolerance of the 16th best (fp64: the 1e-5 gate; fp32 contraction: 5e-3 of the maximum)
        for i in set(dev16.tolist()) ^ set(ref16.tolist()):
            assert abs(rpos[i] - cut) <= 2 * tol, (name, i, rpos[i], cut)
        assert abs(rpos[int(idx[np.argmax(a[idx, 0])])] - r.max()) <= 2 * tol, name
        print("  %s: top-16 set differs from the oracle's in %d members" % (name, len(set(dev16.tolist()) - set(ref16.tolist()))))
    assert set(idx[np.argsort(-a64[idx, 0], kind="stable")[:16]].tolist()) == set(ref16.tolist())
    assert rel64 < 1e-5 and rel32 < 1e-3                      # (measured 1.45e-4: the bound that binds for fp32)


# ---------------------------------------------------------------------------------------------
# (1) a schedule that does not apply on this device (too few CUs for the outputs' reserved share) falls through to the
# single-stream schedule instead of failing with a stale error; (2) a device-side dependency time-out of a gated schedule is not a
# failure of the fit: the attempt is redone on the single-stream schedule, the gated schedules stay off for the context, the event
# is counted.  Both through hooks that exist only in the probes build.
def test_gated_schedule_falls_back(B, probes):
    N, d = 1280, 4                                            # 10 panels, two outputs: the reserved-CU schedule is the default here
    p = R.synthetic_problem(N, d, 2, 32, 8, 91, noise=1e-4)

    def fit(opts, fits=2):
        model = B.multi_outputGP(2, kernel=[_kern(B, "rbf", d, 1.0, p["lengthscales"][j]) for j in range(2)], noise_var=[1e-4, 1e-4], fixed_hyps=True)
        model.incremental = False
        model.set_option("team_fit", 0)                       # (round 4: resident teams are the default at this size; this test is about the launched schedules)
        for k, v in opts:
            model.set_option(k, v)
        for _ in range(fits):                                 # the first factorization of a context is always single-stream
            model.updateModel(p["X"], p["Y"])
        return model

    base = fit([("lookahead", 0)])
    assert base._context().stat("last_schedule") == 0
    L0 = base.get_factor(0)[0]
    gated = fit([])
    if gated._context().stat("cu_masks_ok") == 0:
        pytest.skip("this runtime refuses CU masks: the gated schedules never run")
    assert gated._context().stat("last_schedule") == 2 and gated._context().stat("sched_timeouts") == 0
    assert fit([], fits=1)._context().stat("last_schedule") == 0
    # (1) pretend the device has 16 CUs: 8 reserved would be half of them -> not applicable -> single stream, no error
    small = fit([("force_cu_count", 16)])
    assert small._context().stat("last_schedule") == 0 and small._context().stat("sched_timeouts") == 0
    np.testing.assert_array_equal(small.get_factor(0)[0], L0)
    # (2) a time-out: the fit succeeds and equals the single-stream factor bit for bit; once is forgiven, twice latches the gated
    # schedules off for the context
    t = fit([], fits=1)
    ctx = t._context()
    t.set_option("force_sched_timeout", 1)
    t.updateModel(p["X"], p["Y"])
    assert ctx.stat("sched_timeouts") == 1 and ctx.stat("gated_schedules_off") == 0 and ctx.stat("last_schedule") == 0
    np.testing.assert_array_equal(t.get_factor(0)[0], L0)
    t.updateModel(p["X"], p["Y"])
    assert ctx.stat("last_schedule") == 2 and ctx.stat("sched_timeouts") == 1
    t.set_option("force_sched_timeout", 1)
    t.updateModel(p["X"], [y + 1.0 for y in p["Y"]])
    assert ctx.stat("sched_timeouts") == 2 and ctx.stat("gated_schedules_off") == 1 and ctx.stat("last_schedule") == 0
    t.updateModel(p["X"], p["Y"])
    assert ctx.stat("last_schedule") == 0 and ctx.stat("sched_timeouts") == 2
    np.testing.assert_array_equal(t.get_factor(0)[0], L0)


# The per-rank K* workspace follows the LOCAL shard: a context that only ever scores 8192 candidates holds an
# 8192-column workspace, not the 65 536-column default chunk.
def test_kstar_workspace_is_sized_by_the_local_batch(B):
    N, d, m = 1024, 4, 2
    p = R.synthetic_problem(N, d, m, 8192, 8, 92, noise=1e-4)
    model = _model(B, "rbf", p["X"], p["Y"], p["variances"], p["lengthscales"], p["noise"])
    model.predict(p["Xc"])
    assert model._context().stat("kstar_workspace_bytes") == m * 1024 * 8192 * 8


# ---------------------------------------------------------------------------------------------
# The output-sharded fit on the H x m factorizations of the learning mode: hyper-sample-major outputs
# (gpmodel.py:80-96: one kernel / noise setting per HMC draw and output) dealt to G = 2, 3 simulated ranks, upper tiles packed /
# unpacked the way they travel over RCCL; predictions and the h-averaged acquisition equal the replicated fit's bit for bit.
@pytest.mark.parametrize("G", [2, 3])
def test_sharded_fit_of_hyper_sample_factorizations(B, probes, G):
    F = B._ffi
    lib = F.load()
    rng = np.random.RandomState(17)
    N, d, mo, H, C, S = 300, 3, 2, 3, 200, 16
    M = H * mo
    X = F.f64(rng.uniform(size=(N, d)))
    Yo = rng.normal(size=(mo, N))
    Y = F.f64(np.tile(Yo, (H, 1)))                                    # every hyper-sample sees the same targets
    var = F.f64(rng.uniform(0.5, 2.0, size=M))
    ls = F.f64(rng.uniform(0.3, 0.9, size=(M, d)))
    noise = F.f64(np.full(M, 1e-4))
    Xc = F.f64(rng.uniform(size=(C, d)))
    W = F.f64(rng.normal(size=(S, mo)))
    theta = F.f64([[0.3, -0.1]])
    one = F.f64([1.0])
    res = []
    for sharded in (False, True):
        ctx = F.Context(0)
        ctx.set_option("hyper_samples", H)
        if sharded:
            ctx.set_option("shard_fit_simulate", G)
            ctx.set_option("shard_fit", 1)
        jit, lml = np.zeros(M), np.zeros(M)
        F.check(lib.bocf_fit(ctx.handle, F.dptr(X), F.dptr(Y), N, d, M, F.KERN_RBF, F.dptr(var), F.dptr(ls), F.dptr(noise), 5, F.dptr(jit), F.dptr(lml)),
                "bocf_fit")
        F.check(lib.bocf_set_candidates(ctx.handle, F.dptr(Xc), C), "bocf_set_candidates")
        mean, v = np.empty((M, C)), np.empty((M, C))
        F.check(lib.bocf_predict(ctx.handle, F.ADD_NOISE | F.CLIP, F.dptr(mean), F.dptr(v)), "bocf_predict")
        F.check(lib.bocf_set_mc_samples(ctx.handle, F.dptr(W), S), "bocf_set_mc_samples")
        a_mc, a_lin = np.empty(C), np.empty(C)
        F.check(lib.bocf_acq_mc(ctx.handle, F.ACQ_EI, F.UTIL_NEG_SQ_DIST, None, 0, F.dptr(theta), mo, F.dptr(one), 1, F.dptr(a_mc)), "bocf_acq_mc")
        F.check(lib.bocf_acq_linear(ctx.handle, F.ACQ_EI, F.dptr(theta), F.dptr(one), 1, F.dptr(a_lin)), "bocf_acq_linear")
        res.append((lml, jit, mean, v, a_mc, a_lin))
        ctx.close()
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)
    assert np.isfinite(res[0][4]).all() and res[0][4].max() > 0


# ---------------------------------------------------------------------------------------------
# The device-RESIDENT HMC chain (bocf_hmc: hmc.py:30-69 in one launch) against the lockstep host loop on several outputs at once, with
# a free and a fixed noise, isotropic and ARD lengthscales: same chains, same accept decisions, same final parameters; the golden
# chains of the reference's own hmc.py are in test_device_hmc_chain_golden[resident-*] / test_device_hmc_chain_at_reference_defaults.
@pytest.mark.parametrize("ard,free_noise", [(True, False), (False, True)])
def test_resident_hmc_equals_lockstep_on_several_outputs(B, ard, free_noise):
    from bocf_amd import hyper as H
    rng = np.random.RandomState(5)
    N, d, m, ns, iters, step = 48, 3, 3, 14, 6, 0.03
    X = rng.uniform(size=(N, d))
    Ys = [np.sin(3 * X.dot(rng.normal(size=d)))[:, None] + 0.05 * rng.normal(size=(N, 1)) for _ in range(m)]
    res = []
    for path in ("lockstep", "resident"):
        model = B.multi_outputGP(m, fixed_hyps=False, n_samples=2, ARD=[ard] * m, exact_feval=[not free_noise] * m)
        model._X, model._Y = X, Ys
        model._create_sampler_state()
        outs = model._sampler_outputs
        r2 = np.random.RandomState(9)
        draws = H.LockstepSampler.draw(outs, ns, rng=r2)
        sampler = H.LockstepSampler(outs, model._infer, d, device_hmc=model._device_hmc if path == "resident" else None)
        chains = sampler.hmc([dr[1] for dr in draws], [dr[2] for dr in draws], hmc_iters=iters, stepsize=step)
        res.append((chains, sampler.accepted.copy(), [o.param_array.copy() for o in outs], sampler.n_inferences))
    for j in range(m):
        np.testing.assert_allclose(res[1][0][j], res[0][0][j], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(res[1][2][j], res[0][2][j], rtol=1e-7, atol=1e-10)
    np.testing.assert_array_equal(res[1][1], res[0][1])
    assert res[0][1].sum() > 0 and res[1][3] == 1 + ns * iters


# A factorization that fails inside a trajectory: "raise" stops that output's chain and the binding raises LinAlgError naming it (hmc.py
# lets jitchol's error propagate, linalg.py:71); "reject" rejects the proposal and carries on.  Forced with the diagonal-shift hook
# (probes build): Ky loses definiteness for every parameter setting, whatever the jitter ladder adds.
def test_resident_hmc_failure_modes(B, probes):
    from bocf_amd import hyper as H
    rng = np.random.RandomState(6)
    N, d, m, ns = 32, 2, 2, 5
    X = rng.uniform(size=(N, d))
    Ys = [rng.normal(size=(N, 1)) for _ in range(m)]
    for mode in ("raise", "reject"):
        model = B.multi_outputGP(m, fixed_hyps=False, n_samples=2, exact_feval=[True] * m)
        model._X, model._Y = X, Ys
        model._create_sampler_state()
        outs = model._sampler_outputs
        draws = H.LockstepSampler.draw(outs, ns, rng=np.random.RandomState(3))
        start = [o.param_array.copy() for o in outs]
        sampler = H.LockstepSampler(outs, model._infer, d, device_hmc=model._device_hmc)
        sampler.evaluate()                                    # a good inference first (the lockstep bookkeeping wants a safe point)
        model.set_option("test_diag_shift_1e12", int(50 * 1e12))     # diag(Ky) -= 50: no jitter rung brings it back
        if mode == "raise":
            with pytest.raises(np.linalg.LinAlgError) as e:
                sampler.hmc([dr[1] for dr in draws], [dr[2] for dr in draws], hmc_iters=4, stepsize=0.05, on_failure="raise")
            assert sorted(e.value.outputs) == [0, 1]
        else:
            chains = sampler.hmc([dr[1] for dr in draws], [dr[2] for dr in draws], hmc_iters=4, stepsize=0.05, on_failure="reject")
            assert sampler.accepted.sum() == 0 and (sampler.diverged == ns).all()
            for j in range(m):                                # every draw rejected: the chain never leaves its start
                np.testing.assert_allclose(chains[j], np.tile(start[j][~outs[j].fixed], (ns, 1)), rtol=1e-12)
        model.set_option("test_diag_shift_1e12", 0)


# ---------------------------------------------------------------------------------------------
# Outputs of DIFFERENT kernel families in one model (the reference takes a kernel list, multi_outputGP.py:44-47; bocf_set_kernel_ids):
# every stage that evaluates a kernel -- train build, refined residual, cross-covariance, input gradients, hyper-gradients, append,
# sharded fit, the N <= 128 inference / chain kernels -- against the oracle with one kind per output.
_MIXED = ["matern52", "rbf", "rbf", "matern32", "se", "matern52"]


def _mixed_model(B, kinds, X, Ys, variances, ls, noises):
    d = X.shape[1]
    cls = {"rbf": B.kern.RBF, "se": B.kern.SE, "matern52": B.kern.Matern52, "matern32": B.kern.Matern32}
    kern = [cls[k](d, variance=variances[j], lengthscale=ls[j], ARD=np.size(ls[j]) > 1) for j, k in enumerate(kinds)]
    model = B.multi_outputGP(len(kinds), kernel=kern, noise_var=list(noises), fixed_hyps=True)
    model.updateModel(X, Ys)
    return model


@pytest.mark.parametrize("N", [40, 300])
def test_mixed_kernel_families_fixed_hyps(B, N):
    rng = np.random.RandomState(31 + N)
    d, m, C = 3, len(_MIXED), 77
    X = rng.uniform(size=(N, d))
    Ys = [np.sin(3 * X.dot(rng.normal(size=d)))[:, None] + 0.02 * rng.normal(size=(N, 1)) for _ in range(m)]
    Xc = rng.uniform(-0.1, 1.1, size=(C, d))
    variances = list(rng.uniform(0.5, 2.0, size=m))
    ls = [rng.uniform(0.3, 1.2, size=d) if j % 2 == 0 else np.array([rng.uniform(0.4, 1.0)]) for j in range(m)]
    noises = list(10.0 ** rng.uniform(-5, -3, size=m))
    model = _mixed_model(B, _MIXED, X, Ys, variances, ls, noises)
    ref = R.MultiOutputGPRef(_MIXED, variances, ls, noises)
    ref.updateModel(X, Ys)
    for j in (0, 3, 4):
        np.testing.assert_allclose(model.get_train_kernel(j), R.kern_K(_MIXED[j], X, None, variances[j], ls[j]), rtol=1e-12, atol=1e-14)
    mean, var = model.predict(Xc)
    rm, rv = ref.predict(Xc)
    np.testing.assert_allclose(mean, rm, rtol=1e-6, atol=1e-7)
    assert np.abs(var - rv).max() <= 1e-8 * max(variances)
    np.testing.assert_allclose(model.log_marginal, [o.log_marginal for o in ref.output], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(model.posterior_mean_at_evaluated_points(), ref.posterior_mean_at_evaluated_points(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(model.posterior_mean_gradient(Xc), ref.posterior_mean_gradient(Xc), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(model.posterior_variance_gradient(Xc), ref.posterior_variance_gradient(Xc), rtol=1e-4, atol=1e-7)
    dv, dl, dn = model.log_likelihood_gradients()
    for j, o in enumerate(ref.output):
        rdv, rdl, rdn = o.lml_gradients()
        np.testing.assert_allclose(dv[j], rdv, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(dl[j] if np.size(ls[j]) > 1 else dl[j].sum(), rdl if np.size(ls[j]) > 1 else np.sum(rdl), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(dn[j], rdn, rtol=1e-6, atol=1e-5 * max(1.0, abs(float(rdn))))
    # acquisitions: closed form (+ gradient) and Monte-Carlo
    theta, prob = rng.normal(size=(2, m)), np.array([0.4, 0.6])
    U = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=prob), linear=True)
    a, da = B.maEI(model, None, utility=U)._compute_acq_withGradients(Xc[:9])
    ra, rda = R.ma_acq_with_gradient(*ref.predict(Xc[:9]), ref.posterior_mean_gradient(Xc[:9]), ref.posterior_variance_gradient(Xc[:9]),
                                     ref.posterior_mean_at_evaluated_points(), theta, prob, "EI")
    np.testing.assert_allclose(a, ra, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(da, rda, rtol=1e-4, atol=1e-7)
    W = rng.normal(size=(11, m))
    U2 = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=prob), device="neg_sq_dist")
    acq = B.uEI_noiseless(model, None, utility=U2)
    acq.W_samples = W
    r, _ = R.mc_acq(ref.posterior_mean(Xc), np.sqrt(ref.posterior_variance(Xc)), ref.posterior_mean_at_evaluated_points(), W, "neg_sq_dist", theta, prob, "EI")
    np.testing.assert_allclose(acq._compute_acq(Xc), r, rtol=1e-5, atol=1e-9)
    # one more observation (bocf_append), then new targets only (bocf_update_targets): the kernel families stay with the resident model
    xn = rng.uniform(size=(1, d))
    X2 = np.vstack([X, xn])
    Y2 = [np.vstack([y, [[0.3 * j]]]) for j, y in enumerate(Ys)]
    model.updateModel(X2, Y2)
    ref.updateModel(X2, Y2)
    mean, var = model.predict(Xc)
    rm, rv = ref.predict(Xc)
    np.testing.assert_allclose(mean, rm, rtol=1e-6, atol=1e-7)
    assert np.abs(var - rv).max() <= 1e-8 * max(variances)
    Y3 = [y * 1.5 + 0.1 for y in Y2]
    model.updateModel(X2, Y3)
    ref.updateModel(X2, Y3)
    np.testing.assert_allclose(model.predict(Xc)[0], ref.predict(Xc)[0], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(model.log_marginal, [o.log_marginal for o in ref.output], rtol=1e-9, atol=1e-9)


def test_mixed_kernel_families_sharded_fit(B, probes):
    """Output-sharded fit (simulated ranks inside one process, a hook of the probes build): each share gets its slice of the family list."""
    rng = np.random.RandomState(77)
    N, d, m, C = 260, 2, len(_MIXED), 33
    X = rng.uniform(size=(N, d))
    Ys = [np.cos(2 * X.dot(rng.normal(size=d)))[:, None] for _ in range(m)]
    Xc = rng.uniform(size=(C, d))
    variances, noises = list(rng.uniform(0.5, 2.0, size=m)), [1e-4] * m
    ls = [rng.uniform(0.3, 1.2, size=d) for _ in range(m)]
    ref = R.MultiOutputGPRef(_MIXED, variances, ls, noises)
    ref.updateModel(X, Ys)
    cls = {"rbf": B.kern.RBF, "se": B.kern.SE, "matern52": B.kern.Matern52, "matern32": B.kern.Matern32}
    for G in (2, 4):
        model = B.multi_outputGP(m, kernel=[cls[k](d, variance=variances[j], lengthscale=ls[j], ARD=True) for j, k in enumerate(_MIXED)],
                                 noise_var=noises, fixed_hyps=True)
        model.set_option("shard_fit_simulate", G)
        model.updateModel(X, Ys)
        mean, var = model.predict(Xc)
        rm, rv = ref.predict(Xc)
        np.testing.assert_allclose(mean, rm, rtol=1e-6, atol=1e-7)
        assert np.abs(var - rv).max() <= 1e-8 * max(variances)
        np.testing.assert_allclose(model.log_marginal, [o.log_marginal for o in ref.output], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("N", [24, 100, 180])
def test_mixed_kernel_families_learning(B, N):
    """Learning mode with a kernel list of three families: the device inferences (N <= 128: the fused kernel; above: fit + hyper-gradient
    kernel) agree with the oracle's log-marginal and gradients per output; the resident chain equals the lockstep one; the hyper-samples
    resident after updateModel predict like an oracle model built from the same instances."""
    from bocf_amd import hyper as H
    kinds = ["matern52", "rbf", "matern32"]
    rng = np.random.RandomState(40 + N)
    d, m = 2, 3
    X = rng.uniform(size=(N, d))
    Ys = [np.sin(3 * X.dot(rng.normal(size=d)))[:, None] + 0.05 * rng.normal(size=(N, 1)) for _ in range(m)]
    cls = {"rbf": B.kern.RBF, "matern52": B.kern.Matern52, "matern32": B.kern.Matern32}

    def fresh():
        return B.multi_outputGP(m, kernel=[cls[k](d, variance=1.0 + 0.2 * j, lengthscale=[0.5, 0.8], ARD=True) for j, k in enumerate(kinds)],
                                n_samples=3, fixed_hyps=False)
    model = fresh()
    model._X, model._Y = X, Ys
    model._create_sampler_state()
    params = [(1.3, np.array([0.6, 0.9]), 0.02), (0.8, np.array([0.4, 0.7]), 0.01), (1.1, np.array([0.8, 0.5]), 0.03)]
    lml, dv, dl, dn = model._infer(params)
    for j, (v, l, nz) in enumerate(params):
        o = R.GPFit(kinds[j], X, Ys[j], v, l, nz)
        rdv, rdl, rdn = o.lml_gradients()
        np.testing.assert_allclose(lml[j], o.log_marginal, rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(dv[j], rdv, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(dl[j], rdl, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(dn[j], rdn, rtol=1e-6, atol=1e-5 * max(1.0, abs(float(rdn))))
    if N <= 128:
        res = []
        for path in ("lockstep", "resident"):
            mdl = fresh()
            mdl._X, mdl._Y = X, Ys
            mdl._create_sampler_state()
            outs = mdl._sampler_outputs
            draws = H.LockstepSampler.draw(outs, 8, rng=np.random.RandomState(9))
            sampler = H.LockstepSampler(outs, mdl._infer, d, device_hmc=mdl._device_hmc if path == "resident" else None)
            chains = sampler.hmc([dr[1] for dr in draws], [dr[2] for dr in draws], hmc_iters=5, stepsize=0.03)
            res.append((chains, sampler.accepted.copy()))
        for j in range(m):
            np.testing.assert_allclose(res[1][0][j], res[0][0][j], rtol=1e-7, atol=1e-10)
        np.testing.assert_array_equal(res[1][1], res[0][1])
    model = fresh()
    model.n_burnin, model.subsample_interval, model.leapfrog_steps, model.step_size, model.max_iters = 2, 1, 3, 0.02, 4
    np.random.seed(3)
    model.updateModel(X, Ys)
    Xc = rng.uniform(size=(19, d))
    for h in range(3):
        inst = model._instances[h]
        r = R.MultiOutputGPRef(kinds, [i[0] for i in inst], [i[1] for i in inst], [i[2] for i in inst])
        r.updateModel(X, Ys)
        model.set_hyperparameters(h)
        mean, var = model.predict(Xc)
        rm, rv = r.predict(Xc)
        np.testing.assert_allclose(mean, rm, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(var, rv, rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(model.posterior_variance_gradient(Xc[:4]), r.posterior_variance_gradient(Xc[:4]), rtol=1e-4, atol=1e-7)


# ---------------------------------------------------------------------------------------------
# predict(X, full_cov=True) (multi_outputGP.py:138-149): column 0 of each output's clipped n x n predictive covariance, against the
# reference's own PosteriorExact._raw_predict(full_cov=True) (tests/golden/fullcov.npz) and the oracle; learning mode per hyper-sample.
@pytest.mark.parametrize("tag", ["rbf", "mixed"])
def test_predict_full_cov_golden(B, golden, tag):
    g = golden("fullcov")
    kinds = [str(k) for k in g[tag + "_kinds"]]
    N, d, C, seed = int(g[tag + "_N"]), int(g[tag + "_d"]), int(g[tag + "_C"]), int(g[tag + "_seed"])
    p = R.synthetic_problem(N, d, len(kinds), C, 4, seed)
    model = _mixed_model(B, kinds, p["X"], p["Y"], p["variances"], p["lengthscales"], p["noise"])
    Xc = g[tag + "_Xc"]
    mean, cov = model.predict(Xc, full_cov=True)
    np.testing.assert_allclose(mean, g[tag + "_mean"], rtol=1e-6, atol=1e-7)
    scale = max(p["variances"])
    assert np.abs(cov - g[tag + "_cov0"]).max() <= 1e-8 * scale
    # entry 0 is the predictive variance of the first point, noise included; predict_noiseless ignores the flag as the reference does
    np.testing.assert_allclose(cov[:, 0], model.predict(Xc)[1][:, 0], rtol=1e-9, atol=1e-8 * scale)
    m2, v2 = model.predict_noiseless(Xc, full_cov=True)
    np.testing.assert_array_equal(v2, model.predict_noiseless(Xc)[1])
    # the per-output views hand back an (n, 1) column (gpmodel_fixed_hyps.py:79-87 would give n x n: the wrapper above is its only caller)
    mj, cj = model.output[1].predict(Xc, full_cov=True)
    np.testing.assert_array_equal(cj[:, 0], cov[1])
    # one point, and the device's chunked path (more candidates than one pass holds)
    _, c1 = model.predict(Xc[:1], full_cov=True)
    np.testing.assert_allclose(c1[:, 0], cov[:, 0], rtol=1e-9, atol=1e-8 * scale)
    rng = np.random.RandomState(5)
    big = np.vstack([Xc[:1], rng.uniform(size=(700, d))])
    model.set_option("chunk", 256)
    _, cb = model.predict(big, full_cov=True)
    ref = R.MultiOutputGPRef(kinds, p["variances"], p["lengthscales"], p["noise"])
    ref.updateModel(p["X"], p["Y"])
    assert np.abs(cb - ref.predict(big, full_cov=True)[1]).max() <= 1e-8 * scale


def test_predict_full_cov_learning_mode(B):
    rng = np.random.RandomState(8)
    N, d, m = 40, 2, 2
    X = rng.uniform(size=(N, d))
    Ys = [np.sin(4 * X[:, :1]) + X[:, 1:], np.cos(3 * X[:, 1:]) * X[:, :1]]
    model = B.multi_outputGP(m, n_samples=3, fixed_hyps=False)
    model.n_burnin, model.subsample_interval, model.leapfrog_steps, model.step_size, model.max_iters = 2, 1, 3, 0.02, 4
    np.random.seed(1)
    model.updateModel(X, Ys)
    Xc = rng.uniform(size=(13, d))
    for h in (2, 0):
        inst = model._instances[h]
        r = R.MultiOutputGPRef("se", [i[0] for i in inst], [i[1] for i in inst], [i[2] for i in inst])
        r.updateModel(X, Ys)
        model.set_hyperparameters(h)
        mean, cov = model.predict(Xc, full_cov=True)
        rm, rc = r.predict(Xc, full_cov=True)
        np.testing.assert_allclose(mean, rm, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(cov, rc, rtol=1e-4, atol=1e-8)


# The Monte-Carlo acquisition kernel is specialised on the output count up to m = 8 (acq_mc_m_kernel<M>); more outputs take the generic
# kernel: both against the oracle, m = 1 ... 10, uEI and uPI, with a utility that has an exponential in it.
@pytest.mark.parametrize("m", [1, 2, 7, 8, 9, 10])
def test_mc_acquisition_every_output_count(B, m):
    rng = np.random.RandomState(50 + m)
    N, d, C, S = 40, 3, 130, 192
    X = rng.uniform(size=(N, d))
    Ys = [np.sin(3 * X.dot(rng.normal(size=d)) + j)[:, None] for j in range(m)]
    Xc = rng.uniform(size=(C, d))
    variances, noises = list(rng.uniform(0.5, 1.5, size=m)), [1e-4] * m
    ls = [rng.uniform(0.3, 1.0, size=d) for _ in range(m)]
    model = _model(B, "rbf", X, Ys, variances, ls, noises)
    ref = R.MultiOutputGPRef("rbf", variances, ls, noises)
    ref.updateModel(X, Ys)
    W = rng.normal(size=(S, m))
    theta, prob = 0.3 * rng.normal(size=(2, m)), np.array([0.25, 0.75])
    for util in ("neg_sq_dist", "neg_sum_exp"):
        U = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=prob), device=util)
        for cls, kind in ((B.uEI_noiseless, "EI"), (B.uPI, "PI")):
            acq = cls(model, None, utility=U)
            acq.W_samples = W
            a = acq._compute_acq(Xc)
            r, _ = R.mc_acq(ref.posterior_mean(Xc), np.sqrt(ref.posterior_variance(Xc)), ref.posterior_mean_at_evaluated_points(), W, util, theta, prob, kind)
            np.testing.assert_allclose(a, r, rtol=1e-5, atol=1e-9)
