"""Host logic of hyper-parameter learning on the CPU: the lockstep optimise + HMC driver of
bocf_amd.hyper with the oracle standing in for the device inference, against the oracle's sequential restatement of
gpmodel.py:115-120 / hmc.py:30-69 (itself pinned to the reference's hmc.py by tests/golden/hyper.npz)."""
import numpy as np
import pytest

from bocf_amd import hyper as H
from oracle import cpu_ref as R


def _problem(seed=0, N=24, d=2):
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(N, d))
    Ys = [np.sin(3 * X.sum(1))[:, None], (np.cos(4 * X[:, 0]) + X[:, 1])[:, None] + 0.05 * rng.normal(size=(N, 1)),
          (X[:, 0] * X[:, 1])[:, None]]
    # output 0: ARD, fixed noise; output 1: isotropic, free noise; output 2: ARD, free noise (different P per output)
    spec = [(1.0, [0.8, 0.9], 1e-6, True), (1.3, [0.7], 0.02, False), (0.9, [1.0, 1.1], 0.01, False)]
    return X, Ys, spec


def _oracle_infer(kind, X, Ys):
    def infer(params):
        m, d = len(params), X.shape[1]
        lml, dvar, dls, dnoise = np.empty(m), np.empty(m), np.empty((m, d)), np.empty(m)
        for j, (v, ls, nz) in enumerate(params):
            fit = R.GPFit(kind, X, Ys[j], v, ls, nz)              # ls always (d,): per-dimension gradients
            lml[j] = fit.log_marginal
            dvar[j], dls[j], dnoise[j] = fit.lml_gradients()
        return lml, dvar, dls, dnoise
    return infer


def test_transform_and_prior_match_oracle():
    x = np.array([-30.0, -2.0, 0.5, 7.0, 50.0])
    np.testing.assert_array_equal(H.logexp_f(x), R.logexp_f(x))
    f = H.logexp_f(x)
    np.testing.assert_array_equal(H.logexp_finv(f), R.logexp_finv(f))
    np.testing.assert_array_equal(H.logexp_gradfactor(f, x), R.logexp_gradfactor(f, x))
    np.testing.assert_array_equal(H.logexp_log_jacobian(f[1:]), R.logexp_log_jacobian(f[1:]))
    np.testing.assert_array_equal(H.logexp_log_jacobian_grad(f[1:]), R.logexp_log_jacobian_grad(f[1:]))
    a, b = H.Gamma.from_EV(2., 4.), R.GammaPrior.from_EV(2., 4.)
    t = np.array([0.3, 1.0, 4.0])
    np.testing.assert_allclose(a.lnpdf(t), b.lnpdf(t), rtol=1e-15)
    np.testing.assert_array_equal(a.lnpdf_grad(t), b.lnpdf_grad(t))


def test_lockstep_objective_matches_oracle_model():
    X, Ys, spec = _problem()
    outs = [H.OutputHyper(*s) for s in spec]
    sampler = H.LockstepSampler(outs, _oracle_infer("se", X, Ys), X.shape[1])
    obj, tg = sampler.evaluate()
    for j, s in enumerate(spec):
        ref = R.GPHyperRef("se", X, Ys[j], *s)
        np.testing.assert_allclose(obj[j], ref.objective_function(), rtol=1e-12)
        np.testing.assert_allclose(tg[j], ref._transform_gradients(ref.objective_function_gradients()), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(outs[j].optimizer_array, ref.optimizer_array, rtol=1e-15)
    assert sampler.n_inferences == 1
    sampler.evaluate()
    assert sampler.n_inferences == 1                               # unchanged parameters: no new inference


def test_lockstep_hmc_equals_sequential_reference_order():
    """All outputs advanced together, RNG drawn up front in the reference's order == one output after the other."""
    X, Ys, spec = _problem(1)
    num_samples, iters, step = 9, 4, 0.03
    np.random.seed(77)
    ref_chains, ref_end, ref_eps = [], [], []
    for j, s in enumerate(spec):                                   # gpmodel.py:116-118, output after output
        ref = R.GPHyperRef("se", X, Ys[j], *s)
        ref.param_array[:] = ref.param_array * (1. + np.random.randn(ref.param_array.size) * 0.01)
        ref_chains.append(R.hmc_sample(ref, num_samples, iters, step))
        ref_end.append(ref.param_array.copy())
    after_ref = np.random.rand()
    np.random.seed(77)
    outs = [H.OutputHyper(*s) for s in spec]
    draws = H.LockstepSampler.draw(outs, num_samples)
    after = np.random.rand()
    assert after == after_ref                                      # same RNG consumption
    for o, (eps, _, _) in zip(outs, draws):
        o.param_array[:] = o.param_array * (1. + eps * 0.01)
    sampler = H.LockstepSampler(outs, _oracle_infer("se", X, Ys), X.shape[1])
    chains = sampler.hmc([dr[1] for dr in draws], [dr[2] for dr in draws], hmc_iters=iters, stepsize=step)
    for j in range(3):
        np.testing.assert_allclose(chains[j], ref_chains[j], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(outs[j].param_array, ref_end[j], rtol=1e-8, atol=1e-11)
    assert sampler.n_inferences <= 1 + num_samples * iters          # one batched inference per leapfrog step
    assert np.all(sampler.accepted > 0)


def test_lockstep_optimize_reaches_the_l_bfgs_b_optimum():
    X, Ys, spec = _problem(2)
    outs = [H.OutputHyper(*s) for s in spec]
    sampler = H.LockstepSampler(outs, _oracle_infer("se", X, Ys), X.shape[1])
    f0 = sampler.evaluate()[0].copy()
    info = sampler.optimize(max_iters=200)
    f1 = sampler.evaluate()[0]
    assert np.all(f1 < f0)
    for j, s in enumerate(spec):
        ref = R.GPHyperRef("se", X, Ys[j], *s)
        R.optimize_hyper(ref, 200)
        fr = ref.objective_function()
        assert f1[j] <= fr + 1e-4 * max(1.0, abs(fr)), (j, f1[j], fr)
        if abs(f1[j] - fr) <= 1e-6 * max(1.0, abs(fr)):
            np.testing.assert_allclose(outs[j].param_array, ref.param_array, rtol=2e-2)
    assert sampler.n_inferences < 3 * 200                            # lockstep: far fewer inferences than 3 sequential runs
    assert info["iterations"].max() <= 200


def test_inference_failure_is_infinite_during_optimize_and_fatal_in_hmc():
    X, Ys, spec = _problem(3)
    outs = [H.OutputHyper(*s) for s in spec]
    good = _oracle_infer("se", X, Ys)
    calls = [0]

    def flaky(params):
        calls[0] += 1
        if calls[0] == 3:
            raise np.linalg.LinAlgError("not positive definite, even with jitter.")
        return good(params)
    sampler = H.LockstepSampler(outs, flaky, X.shape[1])
    sampler.optimize(max_iters=30)                                  # paramz Model._objective_grads: +inf, carries on
    assert calls[0] > 3

    def dead(params):
        raise np.linalg.LinAlgError("not positive definite, even with jitter.")
    sampler = H.LockstepSampler([H.OutputHyper(*s) for s in spec], dead, X.shape[1])
    with pytest.raises(np.linalg.LinAlgError):
        sampler.hmc([np.zeros((1, 3)), np.zeros((1, 3)), np.zeros((1, 4))], [np.zeros(1)] * 3, 2, 0.1)


def test_failed_output_is_isolated_and_its_proposal_rejected():
    """One output's trajectory leaves the domain where its Ky factorizes: that output's proposal is rejected, the
    other outputs' chains are exactly what they are without the failure."""
    X, Ys, spec = _problem(4)
    good = _oracle_infer("se", X, Ys)
    num_samples, iters, step = 5, 3, 0.03
    np.random.seed(5)
    outs = [H.OutputHyper(*s) for s in spec]
    draws = H.LockstepSampler.draw(outs, num_samples)
    clean = H.LockstepSampler(outs, good, X.shape[1])
    ref_chains = clean.hmc([d[1] for d in draws], [d[2] for d in draws], iters, step)
    calls = [0]

    def flaky(params):
        calls[0] += 1
        if calls[0] in (6, 7) and params[1][0] != safe_var[0]:
            err = np.linalg.LinAlgError("not positive definite, even with jitter.")
            err.outputs = [1]
            raise err
        return good(params)
    outs2 = [H.OutputHyper(*s) for s in spec]
    sampler = H.LockstepSampler(outs2, flaky, X.shape[1])
    safe_var = [None]

    orig = sampler.evaluate

    def tracking():
        r = orig()
        if not sampler.failed[1]:
            safe_var[0] = sampler._safe[1][0]
        return r
    sampler.evaluate = tracking
    chains = sampler.hmc([d[1] for d in draws], [d[2] for d in draws], iters, step, on_failure="reject")
    np.testing.assert_array_equal(chains[0], ref_chains[0])
    np.testing.assert_array_equal(chains[2], ref_chains[2])
    assert sampler.diverged[1] >= 1 and sampler.diverged[0] == 0 and sampler.diverged[2] == 0
    assert np.all(np.isfinite(chains[1])) and not np.array_equal(chains[1], ref_chains[1])


def test_failed_factorization_raises_by_default_like_the_reference():
    """GPy/util/linalg.py:71 -> hmc.py:62-66 -> gpmodel.py:117-118: the reference lets jitchol's LinAlgError out of updateModel.
    Default on_failure="raise" does the same and names the output; the model is left where the draw started."""
    X, Ys, spec = _problem(4)
    good = _oracle_infer("se", X, Ys)
    outs = [H.OutputHyper(*s) for s in spec]
    np.random.seed(5)
    draws = H.LockstepSampler.draw(outs, 5)
    calls = [0]

    def flaky(params):
        calls[0] += 1
        if calls[0] == 6:
            err = np.linalg.LinAlgError("not positive definite, even with jitter.")
            err.outputs = [1]
            raise err
        return good(params)
    sampler = H.LockstepSampler(outs, flaky, X.shape[1])
    with pytest.raises(np.linalg.LinAlgError) as ei:
        sampler.hmc([d[1] for d in draws], [d[2] for d in draws], 3, 0.03)
    assert ei.value.outputs == [1]
    assert np.all(np.isfinite(outs[1].param_array))
    # an overflowing objective is NOT an exception in the reference either (H_new = nan fails the Metropolis test): rejected
    outs = [H.OutputHyper(*s) for s in spec]

    def overflowing(params):
        lml, dv, dl, dn = good(params)
        if calls[0] > 100:
            lml = lml.copy()
            lml[2] = np.nan
        calls[0] += 1
        return lml, dv, dl, dn
    calls[0] = 100
    sampler = H.LockstepSampler(outs, overflowing, X.shape[1])
    chains = sampler.hmc([d[1] for d in draws], [d[2] for d in draws], 3, 0.03)
    assert sampler.accepted[2] == 0 and np.allclose(chains[2], chains[2][0], rtol=1e-12, atol=0)      # (the Logexp round trip of a restore moves the last bit)


def test_uniform_outputs_vector_path_equals_per_output_path():
    """Outputs with the same parameter count share one set of array operations; same bits as the per-output path,
    including an isotropic kernel (summed lengthscale gradients) and mixed fixed / free noises."""
    rng = np.random.RandomState(9)
    X = rng.uniform(size=(20, 3))
    Ys = [np.sin(3 * X.sum(1))[:, None], (X[:, 0] * X[:, 1] - X[:, 2])[:, None], np.cos(2 * X[:, :1])]
    for spec in ([(1.0, [0.8, 0.9, 1.0], 1e-6, True), (1.3, [0.7, 0.6, 0.9], 0.02, False), (0.9, [1.0, 1.1, 0.5], 1e-3, True)],
                 [(1.0, [0.8], 1e-6, True), (1.3, [0.7], 0.02, False), (0.9, [1.0], 0.01, False)]):
        res = []
        for force_scalar in (False, True):
            outs = [H.OutputHyper(*s) for s in spec]
            sampler = H.LockstepSampler(outs, _oracle_infer("se", X, Ys), 3)
            assert sampler._uniform
            if force_scalar:
                sampler._uniform = False
            np.random.seed(3)
            draws = H.LockstepSampler.draw(outs, 6)
            obj, tg = sampler.evaluate()
            chains = sampler.hmc([dr[1] for dr in draws], [dr[2] for dr in draws], hmc_iters=3, stepsize=0.02)
            res.append((obj.copy(), [t.copy() for t in tg], chains))
        np.testing.assert_array_equal(res[0][0], res[1][0])
        for a, b in zip(res[0][1], res[1][1]):
            np.testing.assert_array_equal(a, b)
        for a, b in zip(res[0][2], res[1][2]):
            np.testing.assert_array_equal(a, b)
