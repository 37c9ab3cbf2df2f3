"""Host logic of the acquisition optimiser on the CPU: the batched box-constrained
quasi-Newton against scipy's L-BFGS-B (what the reference calls, optimizer.py:305) on the same starts, and the
AcquisitionOptimizer flow against a literal restatement of acquisition_optimizer.py:95-154 driven by the oracle."""
import numpy as np
import pytest
import scipy.optimize

from bocf_amd import acquisition_optimizer as AO
from oracle import cpu_ref as R


def _rosen_batch(X):
    X = np.atleast_2d(X)
    f = (100.0 * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1 - X[:, :-1]) ** 2).sum(1)
    g = np.zeros_like(X)
    g[:, :-1] += -400.0 * X[:, :-1] * (X[:, 1:] - X[:, :-1] ** 2) - 2 * (1 - X[:, :-1])
    g[:, 1:] += 200.0 * (X[:, 1:] - X[:, :-1] ** 2)
    return f, g


def test_batched_matches_scipy_on_rosenbrock_with_active_bounds():
    rng = np.random.RandomState(0)
    bounds = [(-1.5, 0.8)] * 2                        # the unconstrained optimum (1, 1) is outside: the bound is active
    X0 = rng.uniform(-1.5, 0.8, size=(12, 2))          # (2-D: one local minimum, so both methods must end at it)
    info = {}
    X, F = AO.lbfgsb_batched(_rosen_batch, X0, bounds, maxiter=500, factr=10.0, pgtol=1e-8, info=info)
    for a in range(12):
        res = scipy.optimize.fmin_l_bfgs_b(lambda x: tuple(v[0] for v in _rosen_batch(x)), X0[a], bounds=bounds, maxiter=500,
                                           factr=10.0, pgtol=1e-8)
        assert F[a] <= res[1] + 1e-6 * max(1.0, abs(res[1])), (a, F[a], res[1])
        if abs(F[a] - res[1]) < 1e-8:                 # same local optimum -> same point
            assert np.allclose(X[a], res[0], atol=2e-4)
    assert info["f_df_calls"] < 400                    # batched: far fewer calls than 12 sequential runs
    assert np.all(X >= -1.5) and np.all(X <= 0.8)


def test_batched_quadratic_exact_and_converged_rows_stop_being_evaluated():
    rng = np.random.RandomState(1)
    Q = rng.randn(5, 5)
    Q = Q @ Q.T + 5 * np.eye(5)
    c = rng.randn(5)
    seen = []

    def f_df(X):
        seen.append(X.shape[0])
        return 0.5 * np.einsum('ad,de,ae->a', X, Q, X) - X @ c, X @ Q - c
    X0 = rng.uniform(-3, 3, size=(8, 5))
    X0[0] = np.linalg.solve(Q, c)                      # already optimal: never evaluated again
    X, F = AO.lbfgsb_batched(f_df, X0, [(-10, 10)] * 5, pgtol=1e-9, factr=1.0)
    assert np.allclose(X, np.linalg.solve(Q, c)[None], atol=1e-6)
    assert seen[0] == 8 and max(seen[1:]) <= 7 and seen[-1] < 7


def test_single_start_signature_and_errors():
    opt = AO.OptLbfgs([(-2.0, 2.0)] * 2)
    x, fx = opt.optimize(np.array([[-1.2, 1.0]]), f=lambda x: _rosen_batch(x)[0][:, None], f_df=_rosen_batch)
    assert x.shape == (1, 2) and fx.shape == (1, 1) and fx[0, 0] < 1e-6
    with pytest.raises(NotImplementedError):
        opt.optimize(np.zeros((1, 2)), f=lambda x: x.sum())
    with pytest.raises(NotImplementedError):
        AO.choose_optimizer('CMA', [(0, 1)])
    assert isinstance(AO.choose_optimizer('lbfgs2', [(0, 1)]), AO.OptLbfgs2)
    sp = AO.Design_space([{'name': 'x', 'type': 'continuous', 'domain': (0, 2), 'dimensionality': 3}])
    assert sp.get_bounds() == [(0.0, 2.0)] * 3 and sp.dimensionality == 3
    with pytest.raises(NotImplementedError):
        AO.Design_space([{'name': 'x', 'type': 'discrete', 'domain': (0, 1, 2)}])


def _oracle_problem(kind):
    N, d, m, S = 14, 2, 3, 64
    p = R.synthetic_problem(N, d, m, 8, S, seed=77)
    model = R.MultiOutputGPRef("rbf", p["variances"], p["lengthscales"], p["noise"])
    model.updateModel(p["X"], p["Y"])
    mu_eval = model.posterior_mean_at_evaluated_points()
    theta = np.full((1, m), 1.0 / m) if kind == "maEI" else np.array([[1.2, -1.0, 1.1]])
    prob = np.ones(1)

    def f_df(Z):
        Z = np.atleast_2d(Z)
        dm, dv = model.posterior_mean_gradient(Z), model.posterior_variance_gradient(Z)
        if kind == "maEI":
            mean, var = model.predict(Z)
            a, g = R.ma_acq_with_gradient(mean, var, dm, dv, mu_eval, theta, prob, "EI")
        else:
            mu, sig = model.posterior_mean(Z), np.sqrt(model.posterior_variance(Z))
            a, g = R.mc_acq_with_gradient(mu, sig, dm, dv, mu_eval, p["W"], "neg_sq_dist", theta, prob)
        return -np.reshape(a, (-1, 1)), -np.reshape(g, Z.shape)

    def f(Z):
        return f_df(Z)[0]
    return f, f_df, d


def _reference_flow(f, f_df, bounds, n_starting, n_anchor):
    """acquisition_optimizer.py:95-154 + anchor_points_generator.py:19-66 + optimizer.py:283-317,463-464, sequential."""
    X = AO.samples_multidimensional_uniform(bounds, n_starting)
    scores = f(X).flatten()
    anchors = X[np.argsort(scores)[:n_anchor]]
    out = []
    for a in anchors:
        res = scipy.optimize.fmin_l_bfgs_b(lambda x: (float(f_df(x)[0][0, 0]), f_df(x)[1][0]), x0=a, bounds=bounds, maxiter=500, factr=1e6)
        out.append((np.atleast_2d(res[0]), f(np.atleast_2d(res[0]))))
    x_min, fx_min = min(out, key=lambda t: t[1][0, 0])
    return anchors, out, x_min, fx_min


@pytest.mark.parametrize("kind", ["maEI", "uEI"])
def test_optimizer_flow_matches_reference_flow_on_the_oracle(kind):
    f, f_df, d = _oracle_problem(kind)
    bounds = [(0.0, 1.0)] * d
    np.random.seed(5)
    anchors, out, x_ref, fx_ref = _reference_flow(f, f_df, bounds, 200, 8)
    np.random.seed(5)
    opt = AO.AcquisitionOptimizer(AO.Design_space(bounds=bounds), n_starting=200, n_anchor=8)
    x_min, fx_min = opt.optimize(f=f, f_df=f_df)
    info = opt.last_info
    # same RNG stream, same selection; candidates with EXACTLY equal scores (acq == 0 plateaus) are ordered arbitrarily by
    # the reference's quicksort and lowest-index-first here, so compare the scores and the untied rows
    ref_scores = f(anchors).flatten()
    assert np.array_equal(info["anchor_points_values"], ref_scores)
    untied = np.array([np.sum(ref_scores == v) == 1 for v in ref_scores])
    assert untied.any() and np.array_equal(info["anchor_points"][untied], anchors[untied])
    assert x_min.shape == (1, d) and np.shape(fx_min) == (1, 1)
    scale = abs(fx_ref[0, 0])
    # every anchor ends at least as low as where it started, and at a point scipy would also call converged
    assert np.all(info["optimized_values"] <= info["anchor_points_values"] + 1e-12)
    # the winner: same optimum as the reference flow (value within L-BFGS-B's own stopping tolerance, same point)
    assert fx_min[0, 0] <= fx_ref[0, 0] + 1e-5 * scale
    if abs(fx_min[0, 0] - fx_ref[0, 0]) <= 1e-5 * scale:
        assert np.allclose(x_min, x_ref, atol=5e-3)
    # per anchor: at least as good as scipy's result from the same start in most rows (different line searches may
    # leave a multimodal start in different basins)
    ref_vals = np.array([o[1][0, 0] for o in out])
    assert np.mean(info["optimized_values"] <= ref_vals + 1e-5 * scale) >= 0.75


def test_baseline_point_wins_when_it_is_best():
    f, f_df, d = _oracle_problem("maEI")
    bounds = [(0.0, 1.0)] * d
    np.random.seed(6)
    opt = AO.AcquisitionOptimizer(bounds, n_starting=50, n_anchor=4)
    x0, f0 = opt.optimize(f=f, f_df=f_df)
    np.random.seed(6)
    x1, f1 = opt.optimize(f=f, f_df=f_df, x_baseline=x0)             # baseline joins the anchors (:127-130)
    assert f1 <= f0 + 1e-12


# ---------------------------------------------------------------------------------------------
# bocf_lbfgsb_batched (C++, what lbfgsb_batched calls) against its NumPy restatement: the same iterates up to the rounding of a dot product
# -- so the same optimum, the same number of callbacks / points / iterations on well-conditioned problems -- for boxed and unbounded
# problems, with maxfun, with rows passed to the callback, with a start that is already optimal, and with a callback that raises.
@pytest.mark.parametrize("boxed", [True, False])
def test_native_batched_lbfgs_equals_numpy_statement(boxed):
    rng = np.random.RandomState(3)
    d = 6
    Q = rng.randn(d, d)
    Q = Q @ Q.T + 0.5 * np.eye(d)
    c = rng.randn(d)

    def f_df(X):
        Z = X - 0.3
        return 0.5 * np.einsum('ad,de,ae->a', Z, Q, Z) - Z @ c + 0.05 * np.sin(3 * Z).sum(1), Z @ Q - c + 0.15 * np.cos(3 * Z)
    bounds = [(0.0, 1.0)] * d if boxed else [(-np.inf, np.inf)] * d
    X0 = rng.uniform(size=(16, d))
    for kw in (dict(), dict(maxfun=7), dict(maxiter=3), dict(factr=1e1, pgtol=1e-10)):
        i0, i1 = {}, {}
        Xn, Fn = AO.lbfgsb_batched_numpy(f_df, X0, bounds, info=i0, **kw)
        Xc, Fc = AO.lbfgsb_batched(f_df, X0, bounds, info=i1, **kw)
        np.testing.assert_allclose(Xc, Xn, rtol=1e-6, atol=1e-6)             # (stopped at the same iteration; the last digits of a dot product differ)
        np.testing.assert_allclose(Fc, Fn, rtol=1e-10, atol=1e-12)
        if "maxfun" in kw or "maxiter" in kw:              # cut off by a count: the same passes, points and iterations
            assert i1["f_df_calls"] == i0["f_df_calls"] and i1["points_evaluated"] == i0["points_evaluated"]
            assert np.array_equal(i1["iterations"], i0["iterations"])
        else:                                              # stopped by the relative-decrease test at ~1e-10: a row may take an iteration more or less
            assert abs(i1["points_evaluated"] - i0["points_evaluated"]) <= 0.15 * i0["points_evaluated"]
            assert np.abs(i1["iterations"] - i0["iterations"]).max() <= 3
        if boxed:
            assert np.all(Xc >= 0.0) and np.all(Xc <= 1.0)


def test_native_batched_lbfgs_rows_optimal_start_and_exceptions():
    seen = []

    def f_df(X, rows):
        seen.append(list(rows))
        return ((X - 0.25) ** 2).sum(1), 2 * (X - 0.25)
    X0 = np.array([[0.9, 0.1], [0.25, 0.25], [0.0, 1.0]])
    X, F = AO.lbfgsb_batched(f_df, X0, [(0, 1)] * 2, with_rows=True)
    assert np.allclose(X, 0.25, atol=1e-6) and np.allclose(F, 0.0, atol=1e-10)
    assert seen[0] == [0, 1, 2] and all(1 not in r for r in seen[1:])          # the optimal start is never evaluated again

    def boom(X):
        raise RuntimeError("device lost")
    with pytest.raises(RuntimeError, match="device lost"):
        AO.lbfgsb_batched(boom, X0, [(0, 1)] * 2)
    # non-finite trial values are rejected by the arc search, a non-finite start is left where it is
    def f_inf(X):
        f = np.where(X[:, 0] > 0.5, np.inf, ((X - 0.2) ** 2).sum(1))
        return f, 2 * (X - 0.2)
    X, F = AO.lbfgsb_batched(f_inf, np.array([[0.4, 0.4], [0.7, 0.7]]), [(0, 1)] * 2)
    assert np.allclose(X[0], 0.2, atol=1e-6) and np.array_equal(X[1], [0.7, 0.7]) and np.isinf(F[1])
