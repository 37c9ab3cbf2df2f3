/*
 * truth_ld.c -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * Extended-precision ("truth") restatement of one exact-GP inference + prediction, in x87 long double (64-bit
 * mantissa, eps = 5.4e-20), so that the fp64 device path AND the fp64 NumPy oracle (oracle/cpu_ref.py) can both be judged
 * against something ~3 orders of magnitude closer to the exact posterior than either of them: at BASELINE configs[2]
 * (N = 4096, cond(Ky) ~ 4e9) fp64 leaves ~1e-7 relative in the variance, long double ~1e-10.
 *
 * Everything -- the ARD scaling of the inputs, the distances, exp / sqrt of the kernel, Ky, the Cholesky factor, the two
 * triangular solves, the sums -- is carried in long double from the fp64 INPUTS (X, Y, hyper-parameters, candidates); only the
 * results are rounded to double.  Follows, line for line:
 *   GPy/kern/src/stationary.py:128-166 (_unscaled_dist / _scaled_dist; here as direct differences, se.py:65-93),
 *   rbf.py:42-43, stationary.py:529-530 (Matern52), :440-441 (Matern32), :168-171 (Kdiag),
 *   exact_gaussian_inference.py:44-51 (Ky = K + (noise + 1e-8) I, alpha = Ky^-1 (Y - mean)),
 *   GPy/util/normalizer.py:57-70 (mean-centring only), posterior.py:299-313 (mean = K* alpha, var = Kxx - sum (L^-1 K*)^2).
 * No jitter ladder: a non-positive pivot returns its 1-based index (the caller treats the case as "no truth available").
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC oracle/truth_ld.c -o oracle/lib/libtruth_ld.so -lm   (oracle/truth.py does it).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef long double ld;

static ld kern_of_r2(int kind, ld variance, ld r2) {
  if (kind <= 1) return variance * expl(-0.5L * r2);                 /* RBF / SE */
  const ld r = sqrtl(r2);
  if (kind == 2) {                                                   /* Matern 5/2 */
    const ld s5r = sqrtl(5.0L) * r;
    return variance * (1.0L + s5r + (5.0L / 3.0L) * r2) * expl(-s5r);
  }
  const ld s3r = sqrtl(3.0L) * r;                                    /* Matern 3/2 */
  return variance * (1.0L + s3r) * expl(-s3r);
}

static ld dot_ld(const ld* a, const ld* b, long n) {
  ld s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  long k = 0;
  for (; k + 4 <= n; k += 4) {
    s0 += a[k] * b[k];
    s1 += a[k + 1] * b[k + 1];
    s2 += a[k + 2] * b[k + 2];
    s3 += a[k + 3] * b[k + 3];
  }
  for (; k < n; ++k) s0 += a[k] * b[k];
  return (s0 + s1) + (s2 + s3);
}

/* In-place lower Cholesky of the row-major n x n matrix A (lower triangle used), left-looking by column blocks of BS:
 * every row of a block column subtracts its dot products with the rows of the block above (parallel over rows), then the
 * diagonal block is factored and the panel below it solved.  Returns 0 or the 1-based index of the first bad pivot. */
#define BS 64
static int chol_ld(ld* A, long n) {
  for (long j0 = 0; j0 < n; j0 += BS) {
    const long jb = (n - j0) < BS ? (n - j0) : BS;
    /* A[i][j0 + c] -= sum_{k < j0} L[i][k] L[j0 + c][k]   for i >= j0 + c */
#pragma omp parallel for schedule(dynamic, 8)
    for (long i = j0; i < n; ++i) {
      const long cmax = (i - j0 + 1) < jb ? (i - j0 + 1) : jb;
      for (long c = 0; c < cmax; ++c) A[i * n + j0 + c] -= dot_ld(A + i * n, A + (j0 + c) * n, j0);
    }
    /* diagonal block */
    for (long c = 0; c < jb; ++c) {
      const long j = j0 + c;
      ld piv = A[j * n + j] - dot_ld(A + j * n + j0, A + j * n + j0, c);
      if (!(piv > 0.0L)) return (int)(j + 1);
      piv = sqrtl(piv);
      A[j * n + j] = piv;
      for (long i = j + 1; i < j0 + jb; ++i) A[i * n + j] = (A[i * n + j] - dot_ld(A + i * n + j0, A + j * n + j0, c)) / piv;
    }
    /* panel below the block: L[i][j0 + c] = (A[i][j0 + c] - sum_{k in block, k < c} L[i][k] L[j0 + c][k]) / L[j0 + c][j0 + c] */
#pragma omp parallel for schedule(static)
    for (long i = j0 + jb; i < n; ++i)
      for (long c = 0; c < jb; ++c) {
        const long j = j0 + c;
        A[i * n + j] = (A[i * n + j] - dot_ld(A + i * n + j0, A + j * n + j0, c)) / A[j * n + j];
      }
  }
  return 0;
}

/*
 * kind: 0 RBF, 1 SE, 2 Matern52, 3 Matern32.  X (N, d), y (N), ls (d), Xc (C, d).
 * mean_out (C): posterior mean incl. the target mean;  var_out (C): RAW variance Kxx - ||L^-1 k*||^2 (no noise, no clip);
 * mu_train_out (N) or NULL: posterior mean at the training inputs;  alpha_out (N) or NULL;  lml_out (1) or NULL.
 * Returns 0, > 0 = first non-positive pivot, < 0 = allocation failure.
 */
int truth_posterior(int kind, const double* X, int N, int d, const double* y, double variance, const double* ls, double noise,
                    const double* Xc, int C, double* mean_out, double* var_out, double* mu_train_out, double* alpha_out,
                    double* lml_out) {
  const long n = N;
  ld* Xs = (ld*)malloc(sizeof(ld) * n * d);
  ld* K = (ld*)malloc(sizeof(ld) * n * n);       /* K(X, X): kept for the mean at the training inputs */
  ld* L = (ld*)malloc(sizeof(ld) * n * n);
  ld* al = (ld*)malloc(sizeof(ld) * n);
  if (!Xs || !K || !L || !al) {
    free(Xs); free(K); free(L); free(al);
    return -1;
  }
  const ld var = (ld)variance;
  for (long i = 0; i < n; ++i)
    for (int q = 0; q < d; ++q) Xs[i * d + q] = (ld)X[i * d + q] / (ld)ls[q];
#pragma omp parallel for schedule(dynamic, 16)
  for (long i = 0; i < n; ++i)
    for (long j = 0; j <= i; ++j) {
      ld r2 = 0;
      for (int q = 0; q < d; ++q) {
        const ld t = Xs[i * d + q] - Xs[j * d + q];
        r2 += t * t;
      }
      const ld k = (i == j) ? var : kern_of_r2(kind, var, r2);
      K[i * n + j] = k;
      K[j * n + i] = k;
    }
  memcpy(L, K, sizeof(ld) * n * n);
  for (long i = 0; i < n; ++i) L[i * n + i] += (ld)noise + 1e-8L;
  const int info = chol_ld(L, n);
  if (info) {
    free(Xs); free(K); free(L); free(al);
    return info;
  }
  /* alpha = Ky^-1 (y - mean) */
  ld ymean = 0;
  for (long i = 0; i < n; ++i) ymean += (ld)y[i];
  ymean /= (ld)n;
  for (long i = 0; i < n; ++i) al[i] = ((ld)y[i] - ymean - dot_ld(L + i * n, al, i)) / L[i * n + i];
  ld quad = 0;                                     /* yc^T Ky^-1 yc = ||L^-1 yc||^2 */
  for (long i = 0; i < n; ++i) quad += al[i] * al[i];
  for (long i = n - 1; i >= 0; --i) {
    ld s = al[i];
    for (long k = i + 1; k < n; ++k) s -= L[k * n + i] * al[k];
    al[i] = s / L[i * n + i];
  }
  if (alpha_out)
    for (long i = 0; i < n; ++i) alpha_out[i] = (double)al[i];
  if (lml_out) {                                   /* exact_gaussian_inference.py:53 */
    ld logdet = 0;
    for (long i = 0; i < n; ++i) logdet += 2.0L * logl(L[i * n + i]);
    *lml_out = (double)(0.5L * (-(ld)n * logl(2.0L * 3.14159265358979323846264338327950288L) - logdet - quad));
  }
  if (mu_train_out) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) mu_train_out[i] = (double)(dot_ld(K + i * n, al, n) + ymean);
  }
  /* candidates */
#pragma omp parallel
  {
    ld* ks = (ld*)malloc(sizeof(ld) * n);
    ld* xc = (ld*)malloc(sizeof(ld) * d);
#pragma omp for schedule(dynamic, 1)
    for (long c = 0; c < C; ++c) {
      for (int q = 0; q < d; ++q) xc[q] = (ld)Xc[c * d + q] / (ld)ls[q];
      for (long i = 0; i < n; ++i) {
        ld r2 = 0;
        for (int q = 0; q < d; ++q) {
          const ld t = Xs[i * d + q] - xc[q];
          r2 += t * t;
        }
        ks[i] = kern_of_r2(kind, var, r2);
      }
      if (mean_out) mean_out[c] = (double)(dot_ld(ks, al, n) + ymean);
      if (var_out) {
        ld ss = 0;
        for (long i = 0; i < n; ++i) {               /* v = L^-1 k*, in place */
          const ld v = (ks[i] - dot_ld(L + i * n, ks, i)) / L[i * n + i];
          ks[i] = v;
          ss += v * v;
        }
        var_out[c] = (double)(var - ss);
      }
    }
    free(ks);
    free(xc);
  }
  free(Xs); free(K); free(L); free(al);
  return 0;
}
