"""Acquisition optimiser with the reference's class surface (GPyOpt/optimization/acquisition_optimizer.py:21-154,
anchor_points_generator.py:9-99, optimizer.py:283-317,425-466), re-shaped for one MI355X:

  reference                                         here
  ------------------------------------------------  ---------------------------------------------------------------
  400 random starts scored by f (one Python loop    n_starting starts (400 by default, 65 536 is one 70 ms device
  per candidate), np.argsort -> 16 anchors          pass) scored in ONE device pass, top-n_anchor chosen on the device
  16 x fmin_l_bfgs_b runs, one anchor after the       all anchors advance together: every iteration is ONE batched
  other (or 4 pathos workers), each iteration one   f_df device pass over the anchors still running; the limited-
  single-point f_df call                            memory quasi-Newton step and the projected line search are
                                                    O(anchors*d) host bookkeeping
  f re-evaluated at every optimum (optimizer.py:    one batched f pass
  464), min over anchors, x_baseline comparison     same

The host consumes np.random exactly as the reference does (one np.random.uniform per input dimension,
random_design.py:67-77), so seeded runs pick identical anchors.  Stopping rules are L-BFGS-B's, with the values the
reference passes (optimizer.py:305: factr=1e6, default pgtol=1e-5, maxiter=500, m=10).  The step itself is a projected
L-BFGS with an Armijo arc search, not a transcription of the Fortran L-BFGS-B (Cauchy point + subspace minimisation +
More-Thuente): iterates differ, local optima found agree (tests/test_optimizer_cpu.py compares both on the same anchors).
"""
import numpy as np

max_objective_anchor_points_logic = "max_objective"
random_design_type = "random"
_EPS = np.finfo(float).eps


class Design_space(object):
    """The slice of GPyOpt/core/task/space.py the path touches, continuous variables only:
    Design_space([{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 4}]) or
    Design_space(bounds=[(lo, hi), ...]).  A GPyOpt Design_space can be passed to AcquisitionOptimizer instead."""

    def __init__(self, space=None, constraints=None, bounds=None):
        if constraints is not None:
            raise NotImplementedError("constraints are outside the device path (SURVEY.md section 8)")
        self.config_space = space
        if bounds is None:
            bounds = []
            for var in space:
                if var.get('type', 'continuous') != 'continuous':
                    raise NotImplementedError("only continuous variables are on the device path")
                bounds += [tuple(var['domain'])] * int(var.get('dimensionality', 1))
        self._bounds = [(float(lo), float(hi)) for lo, hi in bounds]
        self.dimensionality = len(self._bounds)
        self.model_dimensionality = self.dimensionality

    def get_bounds(self):
        return list(self._bounds)

    def get_continuous_bounds(self):
        return list(self._bounds)

    def get_continuous_dims(self):
        return list(range(self.dimensionality))

    def has_continuous(self):
        return True

    def has_constraints(self):
        return False

    def round_optimum(self, x):                    # space.py: continuous variables are returned unchanged
        return np.atleast_2d(x)

    def indicator_constraints(self, x):
        return np.ones((np.atleast_2d(x).shape[0], 1))


def samples_multidimensional_uniform(bounds, points_count):
    """random_design.py:67-77 -- column by column, one np.random.uniform call per dimension (RNG order is part of
    trajectory-level parity)."""
    dim = len(bounds)
    Z_rand = np.zeros(shape=(points_count, dim))
    for k in range(0, dim):
        Z_rand[:, k] = np.random.uniform(low=bounds[k][0], high=bounds[k][1], size=points_count)
    return Z_rand


def _bounds_of(space):
    if hasattr(space, "get_bounds"):
        return [tuple(map(float, b)) for b in space.get_bounds()]
    return [tuple(map(float, b)) for b in space]


class ContextManager(object):
    """acquisition_optimizer.py:333-399 without context variables (none of the experiment scripts sets a context)."""

    def __init__(self, space, context=None):
        if context:
            raise NotImplementedError("context variables are outside the device path")
        self.space = space
        self.noncontext_bounds = _bounds_of(space)
        self.all_index = list(range(len(self.noncontext_bounds)))
        self.noncontext_index = self.all_index[:]
        self.context_index = []

    def _expand_vector(self, x):
        return np.atleast_2d(x)


# ---------------------------------------------------------------------------------------------------------------
#  batched box-constrained limited-memory quasi-Newton
# ---------------------------------------------------------------------------------------------------------------
def lbfgsb_batched(f_df, X0, bounds, maxiter=500, m=10, factr=1e6, pgtol=1e-5, max_ls=20, c1=1e-4, info=None, with_rows=False, maxfun=None):
    """Minimise f from every row of X0 (A, d) at once inside the box `bounds`: bocf_lbfgsb_batched of the library (host arithmetic in
    C++; as NumPy statements -- `lbfgsb_batched_numpy` below, kept as the tests' restatement -- the bookkeeping between two device
    passes cost as much as the pass).  Same arguments and results as the NumPy form; an exception raised by f_df is re-raised."""
    from . import _ffi
    import ctypes
    X = np.ascontiguousarray(np.atleast_2d(X0), dtype=float)
    A, d = X.shape
    lo = np.array([b[0] for b in bounds], dtype=float)
    hi = np.array([b[1] for b in bounds], dtype=float)
    err = []

    def cb(user, Zp, rowsp, n, dd, fp, gp):
        try:
            Z = np.ctypeslib.as_array(Zp, shape=(n, dd)).copy()
            rows = np.ctypeslib.as_array(rowsp, shape=(n,)).copy()
            f, g = f_df(Z, rows) if with_rows else f_df(Z)
            np.ctypeslib.as_array(fp, shape=(n,))[:] = np.asarray(f, dtype=float).reshape(-1)
            np.ctypeslib.as_array(gp, shape=(n, dd))[:] = np.asarray(g, dtype=float).reshape(n, dd)
            return 0
        except BaseException as e:                 # (must not propagate through the C frames)
            err.append(e)
            return 1

    cfun = _ffi.FDF_CALLBACK(cb)
    Xo, Fo = np.empty((A, d)), np.empty(A)
    calls = (ctypes.c_longlong * 2)()
    iters = np.zeros(A, dtype=np.int32)
    rc = _ffi.load().bocf_lbfgsb_batched(ctypes.cast(cfun, ctypes.c_void_p), None, _ffi.dptr(X), A, d, _ffi.dptr(lo), _ffi.dptr(hi), int(maxiter), int(m),
                                         float(factr), float(pgtol), int(max_ls), float(c1), -1 if maxfun is None else int(maxfun), _ffi.dptr(Xo), _ffi.dptr(Fo),
                                         calls, iters.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if err:
        raise err[0]
    if rc != 0:
        raise ValueError("bocf_lbfgsb_batched: bad argument (%d)" % rc)
    if info is not None:
        info.update(f_df_calls=int(calls[0]), points_evaluated=int(calls[1]), iterations=iters.astype(int))
    return Xo, Fo


def lbfgsb_batched_numpy(f_df, X0, bounds, maxiter=500, m=10, factr=1e6, pgtol=1e-5, max_ls=20, c1=1e-4, info=None, with_rows=False, maxfun=None):
    """The algorithm of bocf_lbfgsb_batched as NumPy statements (what tests/test_optimizer_cpu.py checks the library routine against).
    Minimise f from every row of X0 (A, d) at once inside the box `bounds`.

    f_df(X (n, d)) -> (f (n,) or (n, 1), g (n, d)) is called on the rows still running only: one call per trial step,
    i.e. one device pass per iteration for all anchors together.  Stops a row when max|projected gradient| <= pgtol,
    when (f_k - f_{k+1}) / max(|f_k|, |f_{k+1}|, 1) <= factr * eps (both L-BFGS-B's tests), when the arc search
    fails, or after maxiter iterations (or maxfun trial points per row).  Returns (X (A, d), F (A,)); `info`, if a
    dict, receives the counters.  with_rows=True calls f_df(X, rows) with the indices of the rows being evaluated.
    """
    X = np.array(np.atleast_2d(X0), dtype=float)
    A, d = X.shape
    lo = np.array([b[0] for b in bounds], dtype=float)
    hi = np.array([b[1] for b in bounds], dtype=float)
    X = np.minimum(np.maximum(X, lo), hi)
    boxed = bool(np.all(np.isfinite(lo)) and np.all(np.isfinite(hi)))
    calls = [0, 0]

    nfun = np.zeros(A, dtype=int)

    def evaluate(Z, rows):
        f, g = f_df(Z, rows) if with_rows else f_df(Z)
        calls[0] += 1
        calls[1] += Z.shape[0]
        nfun[rows] += 1
        return np.asarray(f, dtype=float).reshape(-1), np.asarray(g, dtype=float).reshape(Z.shape)

    F, G = evaluate(X, np.arange(A))
    S = np.zeros((m, A, d))
    Y = np.zeros((m, A, d))
    RHO = np.zeros((m, A))                     # 0 marks an empty / skipped slot: its two-loop terms vanish
    gamma = np.zeros(A)                        # 0 = no curvature pair yet
    running = np.isfinite(F)
    iters = np.zeros(A, dtype=int)
    head = 0
    for it in range(maxiter):
        # projected gradient: components pushing out of the box are not free
        PG = np.where(((X <= lo) & (G > 0)) | ((X >= hi) & (G < 0)), 0.0, G)
        running &= np.abs(PG).max(axis=1) > pgtol
        if maxfun is not None:
            running &= nfun < maxfun
        if not running.any():
            break
        r = np.flatnonzero(running)
        free = PG[r] != 0.0
        q = PG[r].copy()
        alphas = np.zeros((m, len(r)))
        order = [(head - 1 - j) % m for j in range(m)]           # newest -> oldest
        for j in order:
            alphas[j] = RHO[j, r] * np.einsum('ad,ad->a', S[j, r] * free, q)
            q -= alphas[j][:, None] * (Y[j, r] * free)
        # no curvature pair yet: L-BFGS-B's first move -- B = I and a unit step along the projected steepest-descent path
        # on a fully boxed problem (theta = 1, stp = 1), a step of length 1 (stp = 1 / ||d||) when some variable is unbounded
        g0 = np.where(gamma[r] > 0, gamma[r], 1.0 if boxed else 1.0 / np.maximum(np.sqrt(np.einsum('ad,ad->a', q, q)), 1e-300))
        z = q * g0[:, None]
        for j in reversed(order):
            beta = RHO[j, r] * np.einsum('ad,ad->a', Y[j, r] * free, z)
            z += (S[j, r] * free) * (alphas[j] - beta)[:, None]
        D = -z * free
        slope = np.einsum('ad,ad->a', D, G[r])
        bad = ~(slope < 0)                                        # not a descent direction: steepest descent, drop history
        if bad.any():
            nb = np.ones(int(bad.sum())) if boxed else np.maximum(np.sqrt(np.einsum('ad,ad->a', PG[r][bad], PG[r][bad])), 1e-300)
            D[bad] = -PG[r][bad] / nb[:, None]
            RHO[:, r[bad]] = 0.0
            gamma[r[bad]] = 0.0
        # Armijo search along the projection arc x(t) = P(x + t d)
        t = np.ones(len(r))
        pending = np.ones(len(r), dtype=bool)
        Xn, Fn, Gn = X[r].copy(), F[r].copy(), G[r].copy()
        for _ in range(max_ls):
            p = np.flatnonzero(pending)
            Xt = np.minimum(np.maximum(X[r[p]] + t[p, None] * D[p], lo), hi)
            Ft, Gt = evaluate(Xt, r[p])
            with np.errstate(invalid="ignore", over="ignore"):      # a trial point may have overflowed (f = inf): it is simply rejected
                decrease = np.einsum('ad,ad->a', G[r[p]], Xt - X[r[p]])
                ok = np.isfinite(Ft) & (Ft <= F[r[p]] + c1 * decrease)
            acc = p[ok]
            Xn[acc], Fn[acc], Gn[acc] = Xt[ok], Ft[ok], Gt[ok]
            pending[acc] = False
            if not pending.any():
                break
            rej = p[~ok]
            # safeguarded quadratic interpolation of f along the arc
            with np.errstate(invalid="ignore", over="ignore"):
                num = -decrease[~ok] * t[rej]
                den = 2.0 * (Ft[~ok] - F[r[rej]] - decrease[~ok])
                tq = np.where(np.isfinite(den) & (den > 0), num / np.where(den > 0, den, 1.0), 0.5 * t[rej])
            t[rej] = np.minimum(np.maximum(tq, 0.1 * t[rej]), 0.5 * t[rej])
        failed = r[pending]
        running[failed] = False                                   # arc search failed: keep the current point
        done = r[~pending]
        k = np.flatnonzero(~pending)
        s = Xn[k] - X[done]
        y = Gn[k] - G[done]
        sy = np.einsum('ad,ad->a', s, y)
        yy = np.einsum('ad,ad->a', y, y)
        good = sy > _EPS * yy                                     # L-BFGS-B's curvature test (skip the pair otherwise)
        S[head], Y[head], RHO[head] = 0.0, 0.0, 0.0
        S[head, done[good]] = s[good]
        Y[head, done[good]] = y[good]
        RHO[head, done[good]] = 1.0 / sy[good]
        gamma[done[good]] = sy[good] / yy[good]
        head = (head + 1) % m
        rel = (F[done] - Fn[k]) / np.maximum(np.maximum(np.abs(F[done]), np.abs(Fn[k])), 1.0)
        X[done], F[done], G[done] = Xn[k], Fn[k], Gn[k]
        iters[done] += 1
        running[done[rel <= factr * _EPS]] = False
    if info is not None:
        info.update(f_df_calls=calls[0], points_evaluated=calls[1], iterations=iters.copy())
    return X, F


class Optimizer(object):
    """optimizer.py:10-27."""

    def __init__(self, bounds):
        self.bounds = bounds

    def optimize(self, x0, f=None, df=None, f_df=None):
        raise NotImplementedError("The optimize method is not implemented in the parent class.")


class OptLbfgs(Optimizer):
    """optimizer.py:283-317 (`optimizer='lbfgs'`, what AcquisitionOptimizer.optimize uses): maxiter=500, factr=1e6.
    `optimize_batch` advances many starts together; `optimize` is the reference's one-start signature."""
    factr, pgtol = 1e6, 1e-5

    def __init__(self, bounds, maxiter=500):
        super(OptLbfgs, self).__init__(bounds)
        self.maxiter = maxiter

    def optimize_batch(self, X0, f_df, info=None):
        return lbfgsb_batched(f_df, X0, self.bounds, maxiter=self.maxiter, factr=self.factr, pgtol=self.pgtol, info=info)

    def optimize(self, x0, f=None, df=None, f_df=None):
        if f_df is None and df is not None:
            f_df = lambda x: (f(x), df(x))
        if f_df is None:
            raise NotImplementedError("finite-difference gradients are not on the device path: pass f_df "
                                      "(every bocf_amd acquisition with analytical_gradient_prediction provides it)")
        X, F = self.optimize_batch(np.atleast_2d(x0), f_df)
        return np.atleast_2d(X[0]), np.atleast_2d(F[0])


class OptLbfgs2(OptLbfgs):
    """optimizer.py:319-354 (`inner_optimizer='lbfgs2'`): maxiter=50, factr=1e5, pgtol=1e-15."""
    factr, pgtol = 1e5, 1e-15

    def __init__(self, bounds, maxiter=50):
        super(OptLbfgs2, self).__init__(bounds, maxiter)


def choose_optimizer(optimizer_name, bounds):
    """optimizer.py:583-620; the gradient-free / stochastic choices (DIRECT, CMA, sgd, adam ...) are host-only
    algorithms outside the device path."""
    if optimizer_name == 'lbfgs':
        return OptLbfgs(bounds)
    if optimizer_name == 'lbfgs2':
        return OptLbfgs2(bounds)
    raise NotImplementedError("optimizer %r is outside the device path (lbfgs, lbfgs2 are available)" % (optimizer_name,))


def apply_optimizer(optimizer, x0, f=None, df=None, f_df=None, duplicate_manager=None, context_manager=None, space=None):
    """optimizer.py:425-466 for one start: optimise, then report f at the optimum (`:463-464`)."""
    x0 = np.atleast_2d(x0)
    suggested_x, _ = optimizer.optimize(x0, f, df, f_df)
    return suggested_x, np.atleast_2d(f(suggested_x))


class AnchorPointsGenerator(object):
    """anchor_points_generator.py:9-66."""

    def __init__(self, space, design_type, num_samples):
        if design_type != random_design_type:
            raise NotImplementedError("only the 'random' design (what AcquisitionOptimizer.optimize uses) is provided")
        self.space = space
        self.design_type = design_type
        self.num_samples = num_samples

    def get_anchor_point_scores(self, X):
        raise NotImplementedError("get_anchor_point_scores is not implemented in the parent class.")

    def select(self, scores, num_anchor):
        return np.argsort(scores, kind='stable')[:num_anchor]

    def get(self, num_anchor=8, duplicate_manager=None, unique=False, context_manager=None, get_scores=False):
        X = samples_multidimensional_uniform(_bounds_of(self.space), self.num_samples)
        scores = self.get_anchor_point_scores(X)
        k = min(len(scores), num_anchor)
        idx = self.select(scores, k)
        anchor_points = X[idx, :]
        if get_scores:
            return anchor_points, scores[idx]
        return anchor_points


class ObjectiveAnchorPointsGenerator(AnchorPointsGenerator):
    """anchor_points_generator.py:85-99: scores = objective(X).flatten(), lowest first.  When the objective is the
    `acquisition_function` of a bocf_amd acquisition the scores are still resident on the GPU and the top-k comes from
    the device kernel (ties -> lowest index, which is what a stable argsort gives)."""

    def __init__(self, space, design_type, objective, num_samples=28):
        super(ObjectiveAnchorPointsGenerator, self).__init__(space, design_type, num_samples)
        self.objective = objective

    def get_anchor_point_scores(self, X):
        return self.objective(X).flatten()

    def select(self, scores, num_anchor):
        owner = getattr(self.objective, "__self__", None)
        if owner is not None and hasattr(owner, "select_anchors") and getattr(self.objective, "__name__", "") == "acquisition_function":
            return np.asarray(owner.select_anchors(num_anchor), dtype=np.int64)
        return super(ObjectiveAnchorPointsGenerator, self).select(scores, num_anchor)


class AcquisitionOptimizer(object):
    """acquisition_optimizer.py:21-154.  `optimize(f, df, f_df, duplicate_manager, x_baseline)` -> (x_min (1, d),
    fx_min (1, 1)).  `n_starting` may be raised to tens of thousands: the scoring pass is one device call."""

    def __init__(self, space, optimizer='lbfgs', inner_optimizer='lbfgs2', n_starting=400, n_anchor=16, **kwargs):
        self.space = space
        self.optimizer_name = optimizer
        self.inner_optimizer_name = inner_optimizer
        self.n_starting = n_starting
        self.n_anchor = n_anchor
        self.kwargs = kwargs
        if 'model' in self.kwargs:
            self.model = self.kwargs['model']
        self.type_anchor_points_logic = max_objective_anchor_points_logic
        self.context_manager = ContextManager(space)
        self.optimizer = choose_optimizer(self.optimizer_name, self.context_manager.noncontext_bounds)
        self.inner_optimizer = choose_optimizer(self.inner_optimizer_name, self.context_manager.noncontext_bounds)
        self.last_info = {}

    def optimize(self, f=None, df=None, f_df=None, duplicate_manager=None, x_baseline=None):
        self.f, self.df, self.f_df = f, df, f_df
        if duplicate_manager is not None:
            raise NotImplementedError("duplicate managers are outside the device path (the fork comments them out, "
                                      "anchor_points_generator.py:39-57)")
        if f_df is None:
            raise NotImplementedError("the device path needs f_df (analytical gradients); every bocf_amd acquisition "
                                      "with analytical_gradient_prediction = True passes it (base.py:64-65)")
        self.optimizer = choose_optimizer(self.optimizer_name, self.context_manager.noncontext_bounds)
        generator = ObjectiveAnchorPointsGenerator(self.space, random_design_type, f, self.n_starting)
        anchor_points, anchor_points_values = generator.get(num_anchor=self.n_anchor, duplicate_manager=duplicate_manager,
                                                            context_manager=self.context_manager, get_scores=True)
        f_baseline = None
        if x_baseline is not None:
            x_baseline = np.atleast_2d(x_baseline)
            f_baseline = f(x_baseline)[:, 0]
            anchor_points = np.vstack((anchor_points, x_baseline))
            anchor_points_values = np.concatenate((anchor_points_values, f_baseline))
        info = {}
        Xopt, _ = self.optimizer.optimize_batch(anchor_points, f_df, info=info)
        fx = np.asarray(f(Xopt), dtype=float).reshape(-1)          # optimizer.py:464: f at every optimum, one batch
        best = int(np.argmin(fx))                                   # min(..., key=fx): first of equal minima
        x_min, fx_min = np.atleast_2d(Xopt[best]), np.atleast_2d(fx[best])
        if x_baseline is not None:
            for i in range(x_baseline.shape[0]):
                val = f_baseline[i]
                if val < fx_min:
                    x_min = np.atleast_2d(x_baseline[i, :])
                    fx_min = val
        info.update(anchor_points=anchor_points, anchor_points_values=anchor_points_values, optimized_points=Xopt, optimized_values=fx)
        self.last_info = info
        return x_min, fx_min
