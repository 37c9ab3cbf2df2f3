// Host side of the acquisition optimiser's inner loop: the reference refines its 16 anchor points one after the other with
// scipy.optimize.fmin_l_bfgs_b (GPyOpt/optimization/optimizer.py:283-354, OptLbfgs / OptLbfgs2 -> compiled L-BFGS-B); here ALL starts advance
// together -- one callback, i.e. one device pass over the rows still running, per trial step -- with the same stopping tests (projected
// gradient <= pgtol, relative decrease <= factr * eps, maxiter, maxfun).  The O(A m d) arithmetic between two device passes is this function:
// as NumPy statements it cost as much as the device pass itself (about a hundred small array operations per iteration).
//
// Per row: limited-memory BFGS direction on the free variables (two-loop recursion over the last m curvature pairs; the components whose
// projected gradient vanishes -- at a bound, pushing outwards -- are held), first move as L-BFGS-B makes it (unit step along the projected
// steepest descent on a fully boxed problem, step of length one otherwise), Armijo search along the projection arc with safeguarded
// quadratic interpolation, curvature test s.y > eps y.y before a pair is stored.  Rows never interact: what is batched is the callback.
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/bocf_hip.h"

extern "C" int bocf_lbfgsb_batched(bocf_fdf_callback f_df, void* user, const double* X0, int A, int d, const double* lo, const double* hi,
                                   int maxiter, int m, double factr, double pgtol, int max_ls, double c1, int maxfun, double* X_out,
                                   double* F_out, long long* calls_out, int* iters_out) {
  if (!f_df || !X0 || !lo || !hi || !X_out || !F_out || A < 1 || d < 1 || m < 1 || max_ls < 1 || maxiter < 0) return 1;
  const double eps = std::numeric_limits<double>::epsilon();
  const size_t Ad = (size_t)A * d;
  std::vector<double> X(Ad), F(A), G(Ad), S((size_t)m * Ad, 0.0), Y((size_t)m * Ad, 0.0), RHO((size_t)m * A, 0.0), gamma(A, 0.0);
  std::vector<double> PG(Ad), D(Ad), t(A), Xn(Ad), Fn(A), Gn(Ad), Zb(Ad), fb(A), gb(Ad), alphas(m), q(d), z(d);
  std::vector<int> rows(A), nfun(A, 0), iters(A, 0);
  std::vector<char> running(A), pending(A), isfree(d);
  long long calls = 0, points = 0;
  bool boxed = true;
  for (int k = 0; k < d; ++k) boxed = boxed && std::isfinite(lo[k]) && std::isfinite(hi[k]);
  for (int a = 0; a < A; ++a)
    for (int k = 0; k < d; ++k) {
      const double x = X0[(size_t)a * d + k];
      X[(size_t)a * d + k] = std::fmin(std::fmax(x, lo[k]), hi[k]);
    }
  // one callback over n rows of Zb (rows[] = their indices): f into fb, gradients into gb
  auto evaluate = [&](int n) -> int {
    const int rc = f_df(user, Zb.data(), rows.data(), n, d, fb.data(), gb.data());
    ++calls;
    points += n;
    for (int i = 0; i < n; ++i) ++nfun[rows[i]];
    return rc;
  };
  for (int a = 0; a < A; ++a) rows[a] = a;
  memcpy(Zb.data(), X.data(), sizeof(double) * Ad);
  if (evaluate(A)) return 2;
  for (int a = 0; a < A; ++a) {
    F[a] = fb[a];
    running[a] = std::isfinite(F[a]) ? 1 : 0;
  }
  memcpy(G.data(), gb.data(), sizeof(double) * Ad);
  int head = 0;
  for (int it = 0; it < maxiter; ++it) {
    // projected gradient: components pushing out of the box are not free
    bool any = false;
    for (int a = 0; a < A; ++a) {
      double pgmax = 0.0;
      bool nan = false;
      for (int k = 0; k < d; ++k) {
        const size_t i = (size_t)a * d + k;
        const bool out = (X[i] <= lo[k] && G[i] > 0.0) || (X[i] >= hi[k] && G[i] < 0.0);
        PG[i] = out ? 0.0 : G[i];
        const double v = std::fabs(PG[i]);
        nan = nan || v != v;
        if (v > pgmax) pgmax = v;
      }
      if (nan || !(pgmax > pgtol)) running[a] = 0;         // (a NaN gradient ends the row)
      if (maxfun >= 0 && !(nfun[a] < maxfun)) running[a] = 0;
      any = any || running[a];
    }
    if (!any) break;
    for (int a = 0; a < A; ++a) {
      if (!running[a]) continue;
      const double* pg = &PG[(size_t)a * d];
      const double* g = &G[(size_t)a * d];
      double* dir = &D[(size_t)a * d];
      for (int k = 0; k < d; ++k) {
        isfree[k] = pg[k] != 0.0;
        q[k] = pg[k];
      }
      for (int jj = 0; jj < m; ++jj) {                     // newest -> oldest
        const int j = ((head - 1 - jj) % m + m) % m;
        const double* s = &S[((size_t)j * A + a) * d];
        const double* y = &Y[((size_t)j * A + a) * d];
        double dot = 0.0;
        for (int k = 0; k < d; ++k) dot += (isfree[k] ? s[k] : 0.0) * q[k];
        alphas[j] = RHO[(size_t)j * A + a] * dot;
        for (int k = 0; k < d; ++k) q[k] -= alphas[j] * (isfree[k] ? y[k] : 0.0);
      }
      // no curvature pair yet: L-BFGS-B's first move -- B = I and a unit step along the projected steepest-descent path on a fully boxed
      // problem, a step of length 1 when some variable is unbounded
      double g0;
      if (gamma[a] > 0.0) g0 = gamma[a];
      else if (boxed) g0 = 1.0;
      else {
        double qq = 0.0;
        for (int k = 0; k < d; ++k) qq += q[k] * q[k];
        g0 = 1.0 / std::fmax(std::sqrt(qq), 1e-300);
      }
      for (int k = 0; k < d; ++k) z[k] = q[k] * g0;
      for (int jj = m - 1; jj >= 0; --jj) {                // oldest -> newest
        const int j = ((head - 1 - jj) % m + m) % m;
        const double* s = &S[((size_t)j * A + a) * d];
        const double* y = &Y[((size_t)j * A + a) * d];
        double dot = 0.0;
        for (int k = 0; k < d; ++k) dot += (isfree[k] ? y[k] : 0.0) * z[k];
        const double beta = RHO[(size_t)j * A + a] * dot;
        for (int k = 0; k < d; ++k) z[k] += (isfree[k] ? s[k] : 0.0) * (alphas[j] - beta);
      }
      double slope = 0.0;
      for (int k = 0; k < d; ++k) {
        dir[k] = isfree[k] ? -z[k] : 0.0;
        slope += dir[k] * g[k];
      }
      if (!(slope < 0.0)) {                                // not a descent direction: steepest descent, drop the history
        double nrm = 1.0;
        if (!boxed) {
          double pp = 0.0;
          for (int k = 0; k < d; ++k) pp += pg[k] * pg[k];
          nrm = std::fmax(std::sqrt(pp), 1e-300);
        }
        for (int k = 0; k < d; ++k) dir[k] = -pg[k] / nrm;
        for (int j = 0; j < m; ++j) RHO[(size_t)j * A + a] = 0.0;
        gamma[a] = 0.0;
      }
      t[a] = 1.0;
      pending[a] = 1;
    }
    // Armijo search along the projection arc x(t) = P(x + t d): one callback per trial step for all rows still searching
    for (int ls = 0; ls < max_ls; ++ls) {
      int n = 0;
      for (int a = 0; a < A; ++a) {
        if (!running[a] || !pending[a]) continue;
        for (int k = 0; k < d; ++k) {
          const size_t i = (size_t)a * d + k;
          Zb[(size_t)n * d + k] = std::fmin(std::fmax(X[i] + t[a] * D[i], lo[k]), hi[k]);
        }
        rows[n++] = a;
      }
      if (n == 0) break;
      if (evaluate(n)) return 2;
      for (int i = 0; i < n; ++i) {
        const int a = rows[i];
        const double* xt = &Zb[(size_t)i * d];
        double decrease = 0.0;
        for (int k = 0; k < d; ++k) decrease += G[(size_t)a * d + k] * (xt[k] - X[(size_t)a * d + k]);
        const double ft = fb[i];
        if (std::isfinite(ft) && ft <= F[a] + c1 * decrease) {
          memcpy(&Xn[(size_t)a * d], xt, sizeof(double) * d);
          memcpy(&Gn[(size_t)a * d], &gb[(size_t)i * d], sizeof(double) * d);
          Fn[a] = ft;
          pending[a] = 0;
        } else {                                           // safeguarded quadratic interpolation of f along the arc
          const double num = -decrease * t[a], den = 2.0 * (ft - F[a] - decrease);
          const double tq = (std::isfinite(den) && den > 0.0) ? num / den : 0.5 * t[a];
          t[a] = std::fmin(std::fmax(tq, 0.1 * t[a]), 0.5 * t[a]);
        }
      }
    }
    // the slot the new pairs go to is cleared for every row (a row without a new pair must not keep a stale one there)
    std::fill(S.begin() + (size_t)head * Ad, S.begin() + (size_t)(head + 1) * Ad, 0.0);
    std::fill(Y.begin() + (size_t)head * Ad, Y.begin() + (size_t)(head + 1) * Ad, 0.0);
    std::fill(RHO.begin() + (size_t)head * A, RHO.begin() + (size_t)(head + 1) * A, 0.0);
    for (int a = 0; a < A; ++a) {
      if (!running[a]) continue;
      if (pending[a]) {                                    // arc search failed: keep the current point
        running[a] = 0;
        continue;
      }
      double sy = 0.0, yy = 0.0;
      double* s = &S[((size_t)head * A + a) * d];
      double* y = &Y[((size_t)head * A + a) * d];
      for (int k = 0; k < d; ++k) {
        const size_t i = (size_t)a * d + k;
        s[k] = Xn[i] - X[i];
        y[k] = Gn[i] - G[i];
        sy += s[k] * y[k];
        yy += y[k] * y[k];
      }
      if (sy > eps * yy) {                                 // L-BFGS-B's curvature test (skip the pair otherwise)
        RHO[(size_t)head * A + a] = 1.0 / sy;
        gamma[a] = sy / yy;
      } else {
        for (int k = 0; k < d; ++k) s[k] = y[k] = 0.0;
      }
      const double rel = (F[a] - Fn[a]) / std::fmax(std::fmax(std::fabs(F[a]), std::fabs(Fn[a])), 1.0);
      memcpy(&X[(size_t)a * d], &Xn[(size_t)a * d], sizeof(double) * d);
      memcpy(&G[(size_t)a * d], &Gn[(size_t)a * d], sizeof(double) * d);
      F[a] = Fn[a];
      ++iters[a];
      if (rel <= factr * eps) running[a] = 0;
    }
    head = (head + 1) % m;
  }
  memcpy(X_out, X.data(), sizeof(double) * Ad);
  memcpy(F_out, F.data(), sizeof(double) * A);
  if (calls_out) {
    calls_out[0] = calls;
    calls_out[1] = points;
  }
  if (iters_out) memcpy(iters_out, iters.data(), sizeof(int) * A);
  return 0;
}
