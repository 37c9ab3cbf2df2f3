"""CPU-only checks: the C-ABI library loads and exports every symbol include/bocf_hip.h declares,
the host-side mirror classes behave like the reference's, and nothing computes without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import bocf_amd
    lib = bocf_amd._ffi.load()
    header = open(os.path.join(ROOT, "include", "bocf_hip.h")).read()
    declared = set(re.findall(r"\b(bocf_[a-z_]+)\s*\(", header))
    assert len(declared) >= 23
    for name in declared:
        assert getattr(lib, name) is not None, name
    assert declared == set(bocf_amd._ffi.SIGNATURES), declared ^ set(bocf_amd._ffi.SIGNATURES)
    assert lib.bocf_version() >= 100


def test_option_table_of_the_product_has_no_probe_and_validates_ranges():
    """no option reachable through include/bocf_hip.h may change a result beyond rounding.  The option table is
    enumerable without a GPU: the PRODUCT library holds speed-only options (kind 0) and the documented semantic ones (kind 1), not
    one probe / test hook (kind 2) -- those exist only in libbocf_hip_probes.so (-DBOCF_PROBES) -- and bocf_option_check (what
    bocf_set_option runs first) rejects unknown names and every value outside the table's range."""
    import bocf_amd
    F = bocf_amd._ffi
    lib = F.load()
    table = F.options(lib)
    names = [t[0] for t in table]
    assert len(names) == len(set(names)) >= 20
    semantic = {"predict_f32", "predict_i8", "hyper_samples", "acq_hyper_samples", "best_group", "reuse_data", "skip_mu_train"}
    assert {n for n, _, _, kind, _ in table if kind == 1} == semantic
    assert not [n for n, _, _, kind, _ in table if kind not in (0, 1)]
    hooks = ["kstar_valu_probe", "test_diag_shift_1e12", "shard_fit_simulate", "force_sched_timeout", "force_cu_count", "potrf_scalar"]
    for h in hooks:
        assert h not in names
        assert lib.bocf_option_check(h.encode(), 1) < 0 and b"unknown option" in lib.bocf_last_error()
    for name, lo, hi, kind, what in table:
        assert what
        assert lib.bocf_option_check(name.encode(), lo) == 0 and lib.bocf_option_check(name.encode(), hi) == 0, name
        assert lib.bocf_option_check(name.encode(), lo - 1) < 0 and lib.bocf_option_check(name.encode(), hi + 1) < 0, name
    # values inside a range that select nothing are rejected too; the wrong-result variants of the diagonal-block kernel are not reachable
    # (round 4: the tilings / tile orders / kernels that were kept "for A/B" are gone from the product library: probes build only)
    for name, bad in (("swizzle", 1), ("swizzle", 2), ("swizzle", 3), ("swizzle", 104), ("swizzle", 256), ("swizzle", 257), ("merge_x3", 3),
                      ("lookahead", 1), ("lookahead", 3)):
        assert lib.bocf_option_check(name.encode(), bad) < 0, (name, bad)
    assert lib.bocf_option_check(b"no_such_option", 0) < 0
    # the probes build is the same table plus the hooks
    with F.probes_library() as plib:
        ptable = F.options(plib)
    pnames = [t[0] for t in ptable]
    assert set(names) <= set(pnames) and set(hooks) <= set(pnames)
    assert {n for n, _, _, kind, _ in ptable if kind == 2} == set(hooks)
    for name in ("gemm_waves",):                          # removed outright
        assert name not in names and name not in pnames
    # and the documentation says so
    header = open(os.path.join(ROOT, "include", "bocf_hip.h")).read()
    assert "libbocf_hip_probes.so" in header
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "libbocf_hip_probes.so" in integ and "never change a result beyond rounding" not in integ


def test_header_cites_reference_for_every_compute_entry_point():
    header = open(os.path.join(ROOT, "include", "bocf_hip.h")).read()
    for name in ("bocf_fit", "bocf_predict", "bocf_mean_at_train", "bocf_acq_linear", "bocf_acq_mc", "bocf_select_topk"):
        i = header.index("int " + name)
        comment = header[header.rfind("/*", 0, i): i]
        assert re.search(r"\.py:\d+", comment), name


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bocf_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, (f, "product code must not touch the oracle")
                assert "scipy" not in src, (f, "no CPU numerics in the product path")


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import bocf_amd as B
    model = B.multi_outputGP(1, fixed_hyps=True, device=0)
    with pytest.raises(B._ffi.BocfHipError):
        model.updateModel(np.random.rand(8, 2), [np.random.rand(8, 1)])


def test_kernel_specs_and_duck_typing():
    import bocf_amd as B
    k = B.kern.RBF(3, variance=1.5, lengthscale=[0.2, 0.3, 0.4], ARD=True)
    kid, var, ls = B.kern.kernel_spec(k, 3)
    assert kid == B._ffi.KERN_RBF and var == 1.5 and list(ls) == [0.2, 0.3, 0.4]
    iso = B.kern.SE(3, variance=2., lengthscale=0.3)
    assert list(B.kern.kernel_spec(iso, 3)[2]) == [0.3, 0.3, 0.3]

    class Matern52(object):              # a GPy-like object: class name + variance + lengthscale
        variance = np.array([0.7])
        lengthscale = np.array([1.0, 2.0])
    assert B.kern.kernel_spec(Matern52(), 2)[0] == B._ffi.KERN_MATERN52
    with pytest.raises(NotImplementedError):
        B.kern.kernel_spec(type("Periodic", (), {"variance": 1.0, "lengthscale": 1.0})(), 1)
    with pytest.raises(AssertionError):
        B.kern.RBF(3, lengthscale=[1.0, 2.0], ARD=False)


def test_parameter_distribution_matches_reference_semantics():
    import bocf_amd as B
    sup = np.arange(12.0).reshape(6, 2)
    d = B.ParameterDistribution(continuous=False, support=sup, prob_dist=np.full(6, 1 / 6))
    assert d.use_full_support                      # len(support) < 20  (parameter_distribution.py:18-21)
    big = B.ParameterDistribution(support=np.zeros((20, 2)), prob_dist=np.full(20, 0.05))
    assert not big.use_full_support
    np.random.seed(4)
    s = d.sample(5)
    np.random.seed(4)
    idx = np.random.choice(6, size=5, p=np.full(6, 1 / 6))
    np.testing.assert_array_equal(s, sup[idx, :])


def test_utility_device_spec_and_host_forms():
    import bocf_amd as B
    from oracle import cpu_ref as R
    rng = np.random.RandomState(0)
    y = rng.normal(size=(4, 7))
    th = rng.normal(size=4)
    for name, params, theta in (("linear", None, th), ("neg_sq_dist", None, th), ("neg_sum_exp", None, th),
                                ("neg_exp_cos", [1., 2., 5., 2.], th), ("rosenbrock", None, np.array([1.0]))):
        U = B.Utility(parameter_dist=None, device=name, device_params=params)
        np.testing.assert_allclose(U.eval_func(theta, y), R.utility_eval(name, theta, y, params), rtol=1e-14)
        assert U.device_kind() == R.UTILITY_IDS[name]
    with pytest.raises(ValueError):
        B.Utility(device="nope")
    assert B.Utility(func=lambda t, y: t @ y, linear=True).device == "linear"


def test_acquisition_base_conventions():
    import bocf_amd as B

    class FakeModel(object):
        analytical_gradient_prediction = False
        output_dim = 2

        def number_of_hyps_samples(self):
            return 10

    class A(B.AcquisitionBase):
        analytical_gradient_prediction = True

        def _compute_acq(self, x):
            return np.full((x.shape[0], 1), 3.0)

    a = A(FakeModel(), None, None)
    assert a.analytical_gradient_acq is False           # base.py:22
    np.testing.assert_array_equal(a.acquisition_function(np.zeros((4, 2))), -3.0 * np.ones((4, 1)))   # base.py:40
    c, g = a.cost_withGradients(np.zeros((4, 2)))
    assert c.shape == (4, 1) and g.shape == (4, 2)
    # construction-time RNG touchpoints of uEI_noiseless (uEI_noiseless.py:31,38)
    dist = B.ParameterDistribution(support=np.ones((24, 2)), prob_dist=np.full(24, 1 / 24))
    np.random.seed(9)
    acq = B.uEI_noiseless(FakeModel(), None, utility=B.Utility(parameter_dist=dist, device="neg_sq_dist"))
    np.random.seed(9)
    W = np.random.normal(size=(25, 2))
    np.testing.assert_array_equal(acq.W_samples, W)
    assert acq.utility_params_samples.shape == (10, 2) and acq.n_hyps_samples == 10
    with pytest.raises(TypeError):                       # no CPU fallback for a non-device model
        acq._compute_acq(np.zeros((3, 2)))


def test_model_surface_matches_reference_method_set():
    import bocf_amd as B
    need = ["updateModel", "number_of_hyps_samples", "set_hyperparameters", "get_evaluated_points", "predict", "predict_noiseless",
            "posterior_mean", "posterior_mean_at_evaluated_points", "posterior_variance", "posterior_variance_noiseless",
            "posterior_mean_gradient", "posterior_variance_gradient", "get_model_parameters", "get_model_parameters_names"]
    for n in need:
        assert callable(getattr(B.multi_outputGP, n)), n
    m = B.multi_outputGP(3, fixed_hyps=True, n_samples=7)
    assert m.output_dim == 3 and m.number_of_hyps_samples() == 7 and len(m.output) == 3
    import pickle
    st = pickle.loads(pickle.dumps(m))
    assert st._ctx is None and st.output_dim == 3 and st.analytical_gradient_prediction



def test_oracle_is_only_reachable_from_the_checker_legs():
    """tools/ and examples/ never import the oracle; bench.py only inside cpu_baseline() and the --check branch."""
    import re
    for sub in ("tools", "examples"):
        for f in os.listdir(os.path.join(ROOT, sub)):
            if f.endswith(".py"):
                src = open(os.path.join(ROOT, sub, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), (sub, f)
    src = open(os.path.join(ROOT, "bench.py")).read()
    imports = [m.start() for m in re.finditer(r"from oracle import", src)]
    assert len(imports) == 2
    base = src.index("def cpu_baseline(")
    check = src.index("if a.check and rank == 0:")
    assert any(i > base for i in imports) and any(check < i < check + 200 for i in imports)


def test_synthetic_workload_is_the_oracles():
    from bocf_amd.synthetic import synthetic_problem
    from oracle import cpu_ref as R
    a, b = synthetic_problem(50, 3, 2, 40, 16, 7, noise=1e-4), R.synthetic_problem(50, 3, 2, 40, 16, 7, noise=1e-4)
    for k in ("X", "Xc", "W"):
        np.testing.assert_array_equal(a[k], b[k])
    for ya, yb in zip(a["Y"], b["Y"]):
        np.testing.assert_array_equal(ya, yb)
    for la, lb in zip(a["lengthscales"], b["lengthscales"]):
        np.testing.assert_array_equal(la, lb)
    assert a["noise"] == b["noise"] and a["variances"] == b["variances"]


def test_python_utilities_of_the_experiment_scripts_are_recognised():
    """Utility(func=..., dfunc=..., parameter_dist=..., linear=False) exactly as the reference's scripts build it (no `device`
    argument): the callable is matched to its device form by probing, without touching the global RNG."""
    import bocf_amd as B
    m = 4
    dist_m = B.ParameterDistribution(support=np.array([[0.3, -0.2, 0.5, 0.1]]), prob_dist=np.ones(1))
    dist_1 = B.ParameterDistribution(support=np.array([[1.0]]), prob_dist=np.ones(1))
    cases = [
        (lambda p, y: -np.sum(np.square((y.transpose() - p).transpose()), axis=0), dist_m, "neg_sq_dist"),       # test_1a.py:89-92, test_4a.py:84-87
        (lambda p, y: np.dot(p, y), dist_m, "linear"),                                                            # test_1b.py:89-90
        (lambda p, y: np.sum(-np.exp(y), axis=0), dist_1, "neg_sum_exp"),                                         # test_2a.py:60-62
        (lambda a, y: -sum((a - y[j]) ** 2 + 100 * y[j + 2] ** 2 for j in range(2)), dist_1, "rosenbrock"),       # test_5a.py:48-52 (d - 1 = 2)
    ]
    state = np.random.get_state()[1].copy()
    for func, dist, want in cases:
        U = B.Utility(func=func, dfunc=None, parameter_dist=dist, linear=False)
        assert U.device is None
        assert U.device_kind(m) == B.utility._DEVICE_KINDS[want] and U.device == want
    cw = np.array([1.0, 2.0, 0.5, 3.0])
    U = B.Utility(func=lambda p, y: -np.sum(cw * np.exp(-y / np.pi) * np.cos(np.pi * y)), parameter_dist=dist_1)      # test_3a.py:52-57
    assert U.device_kind(m) == B._ffi.UTIL_NEG_EXP_COS
    np.testing.assert_allclose(U.device_params, cw, rtol=1e-9)
    assert np.array_equal(np.random.get_state()[1], state)                    # the global RNG stream was not consumed
    with pytest.raises(NotImplementedError):
        B.Utility(func=lambda p, y: -np.sum(np.abs(y)), parameter_dist=dist_1).device_kind(m)
    with pytest.raises(NotImplementedError):
        B.Utility(func=lambda p, y: np.dot(p, y) + 1e-6, parameter_dist=dist_m).device_kind(m)      # close is not equal


def test_capi_asan_host_build(tmp_path):
    """Sanitizer row of SURVEY.md section 5: the host side of the C-ABI shim built with AddressSanitizer (device code
    uninstrumented) and driven from C through every entry point's argument-validation path -- null context, null buffers, a
    device that does not exist -- on the CPU.  ASan aborts on any invalid access, LeakSanitizer on any leak."""
    import subprocess
    from bocf_amd import build as b
    lib = b.build_asan_host()
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no clang next to hipcc")
    exe = str(tmp_path / "capi_asan_driver")
    subprocess.check_call([clang, "-fsanitize=address", "-g", "-O1", os.path.join(ROOT, "tests", "capi_asan_driver.c"), "-o", exe,
                           "-L" + os.path.dirname(lib), "-lbocf_hip_asan", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=23")
    r = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = r.stdout.decode("utf-8", "replace")
    assert r.returncode == 0 and "capi asan driver: ok" in out, out[-3000:]


class _MockDeviceModel(object):
    """Duck-typed stand-in for bocf_amd.multi_outputGP on the CPU (the pattern of the reference's own acquisition tests,
    GPyOpt/testing/acquisitions_tests/test_ei_acquisition.py:11-26): posterior queries answered by the oracle."""
    analytical_gradient_prediction = True

    def __init__(self, ref, m):
        self.ref, self.output_dim = ref, m
        self.acq_linear = None                              # marks it as a "device model" for AcquisitionBase._device_model

    def number_of_hyps_samples(self):
        return 1

    def set_hyperparameters(self, h):
        pass

    def __getattr__(self, name):
        return getattr(self.ref, name)


def test_host_fallback_for_arbitrary_utility_cpu():
    """utility.py:37-41 accepts any callable.  One outside the device's closed set is evaluated on the host (loud warning)
    on top of the model's posterior: value and gradient equal the literal loops of uEI_noiseless.py:63-83,138-170."""
    import bocf_amd as B
    from oracle import cpu_ref as R
    N, d, m, C, S = 40, 2, 3, 23, 12
    p = R.synthetic_problem(N, d, m, C, S, 77, noise=1e-4)
    ref = R.MultiOutputGPRef("se", p["variances"], p["lengthscales"], p["noise"])
    ref.updateModel(p["X"], p["Y"])
    model = _MockDeviceModel(ref, m)
    theta = np.array([[0.3, -0.1, 0.2], [0.0, 0.4, -0.3]])
    prob = np.array([0.25, 0.75])
    func = lambda t, y: -np.sum(np.abs((np.asarray(y).T - t).T) ** 1.5, axis=0)            # not one of the device utilities
    dfunc = lambda t, y: -1.5 * np.sign((np.asarray(y).T - t).T) * np.abs((np.asarray(y).T - t).T) ** 0.5
    U = B.Utility(func=func, dfunc=dfunc, parameter_dist=B.ParameterDistribution(support=theta, prob_dist=prob))
    with pytest.raises(NotImplementedError):
        U.device_kind(m)
    for cls, kind in ((B.uEI_noiseless, "EI"), (B.uPI, "PI")):
        acq = cls(model, None, utility=U)
        acq.W_samples = p["W"]
        with pytest.warns(RuntimeWarning, match="HOST"):
            a = acq._compute_acq(p["Xc"])
        # literal triple loop with the user's callable
        mu, sg = ref.posterior_mean(p["Xc"]), np.sqrt(ref.posterior_variance(p["Xc"]))
        fe = ref.posterior_mean_at_evaluated_points()
        want = np.zeros(C)
        for l, th in enumerate(theta):
            best = np.max(func(th, fe))
            for w in p["W"]:
                for i in range(C):
                    v = func(th, mu[:, i] + sg[:, i] * w)
                    want[i] += prob[l] * (max(v - best, 0) if kind == "EI" else float((v - (best + 1e-6)) > 0)) / S
        np.testing.assert_allclose(a[:, 0], want, rtol=1e-12, atol=1e-15)
    acq = B.uEI_noiseless(model, None, utility=U)
    acq.W_samples = p["W"]
    with pytest.warns(RuntimeWarning):
        a, da = acq._compute_acq_withGradients(p["Xc"][:5])
    h = 1e-6
    for q in range(d):
        Xp, Xm = p["Xc"][:5].copy(), p["Xc"][:5].copy()
        Xp[:, q] += h
        Xm[:, q] -= h
        fd = (acq._compute_acq(Xp) - acq._compute_acq(Xm))[:, 0] / (2 * h)
        np.testing.assert_allclose(da[:, q], fd, rtol=5e-4, atol=1e-7 * max(1.0, np.abs(da).max()))
    # a callable that does NOT broadcast over candidates (1-D y only) is evaluated column by column: same numbers
    def one_column(t, y):
        y = np.asarray(y)
        if y.ndim != 1:
            raise ValueError("1-D only")
        return float(-np.sum(np.abs(y - t) ** 1.5))
    U2 = B.Utility(func=one_column, parameter_dist=B.ParameterDistribution(support=theta, prob_dist=prob))
    acq2 = B.uEI_noiseless(model, None, utility=U2)
    acq2.W_samples = p["W"]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_allclose(acq2._compute_acq(p["Xc"]), acq._compute_acq(p["Xc"]), rtol=1e-12, atol=1e-15)
