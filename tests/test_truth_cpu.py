"""CPU checks of the extended-precision truth (oracle/truth_ld.c, test infrastructure): it must itself be pinned before a parity
gate may lean on it -- against 40-digit mpmath arithmetic at a size mpmath can do, and against the fp64 oracle (which is pinned by
the reference's own outputs, tests/test_oracle_golden.py) at a well-conditioned size where fp64 is accurate."""
import numpy as np
import pytest

from oracle import cpu_ref as R
from oracle import truth as T


@pytest.mark.parametrize("kind", ["rbf", "matern52", "matern32"])
def test_truth_matches_mpmath(kind):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    rng = np.random.RandomState(0)
    N, d, C = 20, 3, 4
    X, y, ls, Xc = rng.uniform(size=(N, d)), rng.normal(size=N), np.array([0.7, 1.1, 0.9]), rng.uniform(size=(C, d))
    variance, noise = 1.3, 1e-6
    t = T.posterior_truth(kind, X, y, variance, ls, noise, Xc)

    def k(a, b):
        r2 = sum(((mp.mpf(float(a[q])) - mp.mpf(float(b[q]))) / mp.mpf(float(ls[q]))) ** 2 for q in range(d))
        r = mp.sqrt(r2)
        if kind == "rbf":
            return mp.mpf(variance) * mp.exp(-r2 / 2)
        if kind == "matern52":
            return mp.mpf(variance) * (1 + mp.sqrt(5) * r + mp.mpf(5) / 3 * r2) * mp.exp(-mp.sqrt(5) * r)
        return mp.mpf(variance) * (1 + mp.sqrt(3) * r) * mp.exp(-mp.sqrt(3) * r)

    K = mp.matrix(N, N)
    for i in range(N):
        for j in range(N):
            K[i, j] = k(X[i], X[j]) + ((mp.mpf(noise) + mp.mpf(10) ** -8) if i == j else 0)
    ym = sum(mp.mpf(float(v)) for v in y) / N
    al = mp.lu_solve(K, mp.matrix([mp.mpf(float(v)) - ym for v in y]))
    for c in range(C):
        ks = mp.matrix([k(X[i], Xc[c]) for i in range(N)])
        mean = sum(ks[i] * al[i] for i in range(N)) + ym
        w = mp.lu_solve(K, ks)
        var = mp.mpf(variance) - sum(ks[i] * w[i] for i in range(N))
        assert abs(float(mean - mp.mpf(float(t["mean"][c])))) < 1e-12 * max(1.0, abs(float(mean)))
        assert abs(float(var - mp.mpf(float(t["raw_var"][c])))) < 1e-13 * variance
    mu_tr = [float(sum(k(X[i], X[j]) * al[j] for j in range(N)) + ym) for i in range(N)]
    np.testing.assert_allclose(t["mu_train"], mu_tr, rtol=0, atol=1e-12)


def test_truth_matches_oracle_where_fp64_is_accurate():
    p = R.synthetic_problem(300, 4, 2, 50, 8, 5, noise=1e-2)
    tru = T.model_truth("se", p["X"], p["Y"], p["variances"], p["lengthscales"], p["noise"], p["Xc"])
    ref = R.MultiOutputGPRef("se", p["variances"], p["lengthscales"], p["noise"])
    ref.updateModel(p["X"], p["Y"])
    rm, rv = ref.predict(p["Xc"])
    np.testing.assert_allclose(rm, tru["mean"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(rv, tru["var"], rtol=1e-9)
    np.testing.assert_allclose(ref.posterior_mean_at_evaluated_points(), tru["mu_train"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose([g.log_marginal for g in ref.output], tru["lml"], rtol=1e-11)


def test_gate_arithmetic():
    ok, e_dev, e_orc, bound = T.gate([1.0, 2.0 + 3e-6], [1.0, 2.0 + 1e-6], [1.0, 2.0], scale=2.0)
    assert ok and abs(e_dev - 3e-6) < 1e-12 and abs(e_orc - 1e-6) < 1e-12 and abs(bound - 2e-5) < 1e-12
    assert not T.gate([2.0 + 1e-4], [2.0 + 1e-6], [2.0], scale=2.0)[0]
    assert T.gate([2.0 + 1e-4], [2.0 + 3e-5], [2.0], scale=2.0)[0]      # the oracle itself is 3e-5 off: 4 x that is the bound
