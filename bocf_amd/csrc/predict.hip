// Predict-side kernels: cross kernel K* = K(X, X*) with the posterior mean fused in, and the
// small finalisation kernels.  The N^2 C contraction itself is gemm_f64.hip.
#include "bocf_internal.h"
#include "../../include/bocf_hip.h"

__device__ __forceinline__ double kern_of_r2_p(int kernel_id, double variance, double r2) {
  if (kernel_id <= 1) return variance * bocf_exp_nonpos(-0.5 * r2);
  const double r = sqrt(r2);
  if (kernel_id == 2) {
    const double s5r = 2.23606797749978969641 * r;
    return variance * (1.0 + s5r + (5.0 / 3.0) * r2) * bocf_exp_nonpos(-s5r);
  }
  const double s3r = 1.73205080756887729353 * r;
  return variance * (1.0 + s3r) * bocf_exp_nonpos(-s3r);
}

// One thread per PAIR of adjacent candidate columns, looping over a slice of the training points.  The training point is
// wave-uniform (scalar loads), the two candidates' scaled coordinates live in registers, the store K*[kk][c..c+1] is one
// 16-B-per-lane instruction (a workgroup writes a contiguous 4 KiB row segment; 8-B stores reach only 0.5-0.7 of the 16-B
// store rate on gfx950), the two independent exp chains hide each other's latency, and the posterior mean K(x*,X) alpha
// (posterior.py:299-305) accumulates on the fly in a fixed order.  Per element the arithmetic is that of the one-column form.
typedef double v2d_p __attribute__((ext_vector_type(2)));
typedef float v2f_p __attribute__((ext_vector_type(2)));
template <int D, int KID, int STORE>
__global__ __launch_bounds__(256) void cross_kernel(const double* __restrict__ Xs, long strideXs, int N, int Np, int kernel_id_unused,
                                                    const KernHyp* __restrict__ hyp, const double* __restrict__ Xc, int c0, int Cn,
                                                    const double* __restrict__ alpha, double* __restrict__ Kstar, long ldk, long strideK,
                                                    double* __restrict__ meanpart, double* __restrict__ meanlo, int nsplit, int Cpad,
                                                    int store_k, int jbase, int mtot) {
  const int j = blockIdx.z;
  const int split = blockIdx.y;
  const int c = (blockIdx.x * 256 + threadIdx.x) * 2;   // first column of the pair inside this chunk (Cpad is even)
  if (c >= Cpad) return;
  const bool valid0 = c < Cn, valid1 = c + 1 < Cn;
  const KernHyp h = hyp[j];
  double xa[D], xb[D];
#pragma unroll
  for (int q = 0; q < D; ++q) {
    xa[q] = valid0 ? Xc[(long)(c0 + c) * D + q] / h.ls[q] : 0.0;
    xb[q] = valid1 ? Xc[(long)(c0 + c + 1) * D + q] / h.ls[q] : 0.0;
  }
  // rows are processed in blocks of 128 and every block writes its own partial mean, so the
  // result of a candidate never depends on how many splits / which batch it was evaluated in
  const int nblk = Np / BOCF_TILE;
  const int bps = (nblk + nsplit - 1) / nsplit;
  const int b0 = split * bps;
  int b1 = b0 + bps;
  if (b1 > nblk) b1 = nblk;
  const double* __restrict__ X = Xs + (long)j * strideXs;
  const double* __restrict__ al = alpha + (long)j * Np;
  double* __restrict__ Kj = Kstar + (long)j * strideK;
  float* __restrict__ Kf = reinterpret_cast<float*>(Kstar) + (long)j * strideK;   // fp32 store variant
  for (int blk = b0; blk < b1; ++blk) {
    double mean0 = 0.0, mean1 = 0.0, lo0 = 0.0, lo1 = 0.0;
    const int kbeg = blk * BOCF_TILE;
    for (int kk = kbeg; kk < kbeg + BOCF_TILE; ++kk) {
      double v0 = 0.0, v1 = 0.0;
      if (kk < N) {
        double r0 = 0.0, r1 = 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) {
          const double xq = X[(long)kk * D + q];
          const double d0 = xq - xa[q], d1 = xq - xb[q];
          r0 += d0 * d0;
          r1 += d1 * d1;
        }
        v0 = kern_of_r2_p(KID, h.variance, r0);
        v1 = kern_of_r2_p(KID, h.variance, r1);
        const double a = al[kk];
        dd_fma_acc(mean0, lo0, v0, a);
        dd_fma_acc(mean1, lo1, v1, a);
      }
      // (non-temporal: K* is gigabytes, read once by the contraction; with plain stores this loop runs a quarter slower -- tools/hbm_kernel_probe.hip)
      if (STORE == 1) __builtin_nontemporal_store((v2d_p){valid0 ? v0 : 0.0, valid1 ? v1 : 0.0}, reinterpret_cast<v2d_p*>(Kj + (long)kk * ldk + c));
      else if (STORE == 2) __builtin_nontemporal_store((v2f_p){valid0 ? (float)v0 : 0.f, valid1 ? (float)v1 : 0.f}, reinterpret_cast<v2f_p*>(Kf + (long)kk * ldk + c));
    }
    // (partial means are laid out [block][output of the WHOLE model][column]: a launch that covers a run of outputs writes its slots)
    *reinterpret_cast<v2d_p*>(meanpart + ((long)blk * mtot + jbase + j) * Cpad + c) = (v2d_p){mean0, mean1};
    *reinterpret_cast<v2d_p*>(meanlo + ((long)blk * mtot + jbase + j) * Cpad + c) = (v2d_p){lo0, lo1};
  }
}

template <int D>
static void launch_cross_d(const double* Xs, long strideXs, int N, int Np, int kernel_id, const KernHyp* hyp, const double* Xc, int c0,
                           int Cn, int Cpad, const double* alpha, double* Kstar, long ldk, long strideK, double* meanpart, double* meanlo,
                           int nsplit, int m, int store_k, hipStream_t s, int jbase, int mtot) {
  dim3 grid((unsigned)((Cpad + 511) / 512), (unsigned)nsplit, (unsigned)m);   // 256 threads x 2 columns
  const int kid = kernel_id <= 1 ? 0 : kernel_id;
#define LAUNCH(KID, ST)                                                                                                        \
  BOCF_LAUNCH((cross_kernel<D, KID, ST>), grid, dim3(256), 0, s, Xs, strideXs, N, Np, kernel_id, hyp, Xc, c0, Cn, alpha, \
                     Kstar, ldk, strideK, meanpart, meanlo, nsplit, Cpad, store_k, jbase, mtot)
#define BYSTORE(KID)                          \
  if (store_k == 0) LAUNCH(KID, 0);           \
  else if (store_k == 1) LAUNCH(KID, 1);      \
  else LAUNCH(KID, 2)
  if (kid == 0) { BYSTORE(0); }
  else if (kid == 2) { BYSTORE(2); }
  else { BYSTORE(3); }
#undef BYSTORE
#undef LAUNCH
}

static void launch_cross_kernel_run(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* Xc, int c0, int Cn,
                                    int Cpad, const double* alpha, double* Kstar, long ldk, long strideK, double* meanpart, double* meanlo, int nsplit,
                                    int m, int store_k, hipStream_t s, int jbase, int mtot);
void launch_cross_kernel(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* Xc,
                         int c0, int Cn, int Cpad, const double* alpha, double* Kstar, long ldk, long strideK, double* meanpart,
                         double* meanlo, int nsplit, int m, int store_k, hipStream_t s, const int* kids) {
  int jbase = 0, mtot = m;
  if (kids) {
    const int* none = nullptr;
    bocf_family_runs(kernel_id, kids, m, [&](int j0, int mr, int kid_) {
      // K* of a run starts at its first output (fp64 or fp32 storage: strideK counts ELEMENTS of the stored type)
      double* kst = !Kstar ? nullptr : (store_k == 2 ? reinterpret_cast<double*>(reinterpret_cast<float*>(Kstar) + (long)j0 * strideK) : Kstar + (long)j0 * strideK);
      launch_cross_kernel_run(Xs + (long)j0 * strideXs, strideXs, N, Np, d, kid_, hyp + j0, Xc, c0, Cn, Cpad, alpha + (long)j0 * Np, kst, ldk, strideK,
                              meanpart, meanlo, nsplit, mr, store_k, s, j0, m);
    });
    (void)none;
    return;
  }
  launch_cross_kernel_run(Xs, strideXs, N, Np, d, kernel_id, hyp, Xc, c0, Cn, Cpad, alpha, Kstar, ldk, strideK, meanpart, meanlo, nsplit, m, store_k, s,
                          jbase, mtot);
}

static void launch_cross_kernel_run(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* Xc, int c0, int Cn,
                             int Cpad, const double* alpha, double* Kstar, long ldk, long strideK, double* meanpart, double* meanlo, int nsplit,
                             int m, int store_k, hipStream_t s, int jbase, int mtot) {
#define CASE(D)                                                                                                                  \
  case D:                                                                                                                        \
    launch_cross_d<D>(Xs, strideXs, N, Np, kernel_id, hyp, Xc, c0, Cn, Cpad, alpha, Kstar, ldk, strideK, meanpart, meanlo, nsplit, m, \
                      store_k, s, jbase, mtot);                                                                                               \
    break;
  switch (d) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
    CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29)
    CASE(30) CASE(31) CASE(32)
    default: break;
  }
#undef CASE
}

// mean[j][c0 + c] = sum_blk meanpart[blk][j][c] + ymean_j   (gp.py:393-399, normalizer.py:67-68)
__global__ void finalize_mean_kernel(const double* __restrict__ meanpart, const double* __restrict__ meanlo, int nsplit, int Cpad,
                                     const KernHyp* __restrict__ hyp, double* __restrict__ mean, long ldmean, int c0, int Cn, int m) {
  const int j = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cn) return;
  double s = 0.0, lo = 0.0;                              // the per-block pairs are summed as pairs, in block order
  for (int sp = 0; sp < nsplit; ++sp) dd_add_acc(s, lo, meanpart[((long)sp * m + j) * Cpad + c], meanlo[((long)sp * m + j) * Cpad + c]);
  mean[(long)j * ldmean + c0 + c] = (s + lo) + hyp[j].ymean;
}

void launch_finalize_mean(const double* meanpart, const double* meanlo, int nsplit, int Cpad, const KernHyp* hyp, double* mean, long ldmean,
                          int c0, int Cn, int m, hipStream_t s) {
  if (Cn == 0) return;
  BOCF_LAUNCH(finalize_mean_kernel, dim3((unsigned)((Cn + 255) / 256), (unsigned)m), dim3(256), 0, s, meanpart, meanlo, nsplit, Cpad, hyp,
                     mean, ldmean, c0, Cn, m);
}

// var[j][c0 + c] = sigma_f^2 - sum_rt sumsq[j][rt][c]  [+ noise]  [clip 1e-10]
// (posterior.py:309-313; gaussian.py:100-101,110-111; gpmodel.py:147,174,183)
// mp != nullptr: the same launch also finishes the means (finalize_mean_kernel's arithmetic: one launch fewer behind the contraction)
__global__ void finalize_var_kernel(const double* __restrict__ sumsq, int nrt, int Cpad, const KernHyp* __restrict__ hyp, int flags,
                                    double* __restrict__ var, long ldvar, int c0, int Cn, const double* __restrict__ mp,
                                    const double* __restrict__ mlo, int nsplit, double* __restrict__ mean, int m) {
  const int j = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cn) return;
  double s = 0.0;
  for (int rt = 0; rt < nrt; ++rt) s += sumsq[((long)j * nrt + rt) * Cpad + c];
  double v = hyp[j].variance - s;
  if (flags & BOCF_ADD_NOISE) v += hyp[j].noise;
  if ((flags & BOCF_CLIP) && !(v >= 1e-10)) v = 1e-10;
  var[(long)j * ldvar + c0 + c] = v;
  if (mp) {
    double ms = 0.0, lo = 0.0;                           // the per-block pairs are summed as pairs, in block order
    for (int sp = 0; sp < nsplit; ++sp) dd_add_acc(ms, lo, mp[((long)sp * m + j) * Cpad + c], mlo[((long)sp * m + j) * Cpad + c]);
    mean[(long)j * ldvar + c0 + c] = (ms + lo) + hyp[j].ymean;
  }
}

// Column 0 of the predictive covariance (multi_outputGP.py:146-148 keeps tmp2[:, 0] of posterior.py:274-283's n x n matrix):
// cov0[j][c] = k_j(x_c, x_0) - (t[j][c] - ymean_j)  [+ noise_j at c == 0]  [clipped at 1e-10: gpmodel_fixed_hyps.py:86 clips every entry],
// t = the mean-shaped pass K(x_c, X) w + ymean with w = Ky^-1 k(X, x_0) in the place of alpha.
__global__ void cov_column_kernel(const double* __restrict__ Xc, int C, int d, int kernel_id,
                                  const KernHyp* __restrict__ hyp, const double* __restrict__ t, long ldt, int flags,
                                  double* __restrict__ cov, long ldcov) {
  const int j = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const KernHyp h = hyp[j];
  double r2 = 0.0;
  for (int q = 0; q < d; ++q) {
    const double dq = Xc[(long)c * d + q] / h.ls[q] - Xc[q] / h.ls[q];
    r2 += dq * dq;
  }
  double v = kern_of_r2_p(kernel_id, h.variance, r2) - (t[(long)j * ldt + c] - h.ymean);
  if (c == 0 && (flags & BOCF_ADD_NOISE)) v += h.noise;
  if ((flags & BOCF_CLIP) && !(v >= 1e-10)) v = 1e-10;
  cov[(long)j * ldcov + c] = v;
}

void launch_cov_column(const double* Xc, int C, int d, int kernel_id, const KernHyp* hyp, const double* t, long ldt, int flags, double* cov,
                       long ldcov, int m, hipStream_t s, const int* kids) {
  if (C == 0) return;
  bocf_family_runs(kernel_id, kids, m, [&](int j0, int mr, int kid) {
    BOCF_LAUNCH(cov_column_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)mr), dim3(256), 0, s, Xc, C, d, kid, hyp + j0, t + (long)j0 * ldt, ldt,
                flags, cov + (long)j0 * ldcov, ldcov);
  });
}

// the same with MANY partials per candidate (the small path: one per 16-row tile): a wave per (candidate, output), lane l adds tiles
// l, l + 64, ... in order, then a fixed butterfly -- the sum of a candidate does not depend on how many candidates share the launch
// (mp != nullptr: lane 0 also finishes the candidate's mean -- the serial pair sum of finalize_mean_kernel over the nsplit row blocks)
__global__ void finalize_var_wave_kernel(const double* __restrict__ sumsq, int nrt, int Cpad, const KernHyp* __restrict__ hyp, int flags,
                                         double* __restrict__ var, long ldvar, int c0, const double* __restrict__ mp, const double* __restrict__ mlo,
                                         int nsplit, double* __restrict__ mean, int m) {
  const int j = blockIdx.y, c = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int rt = lane; rt < nrt; rt += 64) s += sumsq[((long)j * nrt + rt) * Cpad + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) {
    double v = hyp[j].variance - s;
    if (flags & BOCF_ADD_NOISE) v += hyp[j].noise;
    if ((flags & BOCF_CLIP) && !(v >= 1e-10)) v = 1e-10;
    var[(long)j * ldvar + c0 + c] = v;
    if (mp) {
      double ms = 0.0, lo = 0.0;
      for (int sp = 0; sp < nsplit; ++sp) dd_add_acc(ms, lo, mp[((long)sp * m + j) * Cpad + c], mlo[((long)sp * m + j) * Cpad + c]);
      mean[(long)j * ldvar + c0 + c] = (ms + lo) + hyp[j].ymean;
    }
  }
}

void launch_finalize_var(const double* sumsq, int nrt, int Cpad, const KernHyp* hyp, int flags, double* var, long ldvar, int c0, int Cn, int m, hipStream_t s,
                         const double* meanpart, const double* meanlo, int nsplit, double* mean) {
  if (Cn == 0) return;
  if (nrt > 32 && Cn <= 64) {                              // (the small path: many 16-row partials, few candidates -- a wave per candidate)
    BOCF_LAUNCH(finalize_var_wave_kernel, dim3((unsigned)Cn, (unsigned)m), dim3(64), 0, s, sumsq, nrt, Cpad, hyp, flags, var, ldvar, c0, meanpart, meanlo,
                nsplit, mean, m);
    return;
  }
  BOCF_LAUNCH(finalize_var_kernel, dim3((unsigned)((Cn + 255) / 256), (unsigned)m), dim3(256), 0, s, sumsq, nrt, Cpad, hyp, flags,
                     var, ldvar, c0, Cn, meanpart, meanlo, nsplit, mean, m);
}

// ---------------------------------------------------------------------------------------------
// Input gradients of the posterior:
//   d mu / dx   = sum_i alpha_i      dk(x, X_i)/dx        GP.posterior_mean_gradient      gp.py:438-461
//   d var / dx  = sum_i -2 w_i       dk(x, X_i)/dx        GP.posterior_variance_gradient  gp.py:464-490, w = Ky^-1 k(X, x)
// with dk/dx_q = f(r) (x_q - X_iq) / l_q^2, f = invdist * dK_dr (stationary.py:312-331, se.py:135-148):
//   RBF/SE  -k(r);   Matern52  -(5/3) s2 (1 + sqrt5 r) e^{-sqrt5 r};   Matern32  -3 s2 e^{-sqrt3 r}.
// One workgroup per (candidate, output); lanes stride the training points; fixed-order reduction.
__device__ __forceinline__ double kern_dfac(int kernel_id, double variance, double r2) {
  if (kernel_id <= 1) return -variance * bocf_exp_nonpos(-0.5 * r2);
  const double r = sqrt(r2);
  if (kernel_id == 2) {
    const double s5r = 2.23606797749978969641 * r;
    return -(5.0 / 3.0) * variance * (1.0 + s5r) * bocf_exp_nonpos(-s5r);
  }
  return -3.0 * variance * bocf_exp_nonpos(-1.73205080756887729353 * r);
}

template <int D>
__global__ __launch_bounds__(256) void grad_kernel(const double* __restrict__ Xs, long strideXs, int N, int Np, int kernel_id,
                                                   const KernHyp* __restrict__ hyp, const double* __restrict__ Xc, int c0,
                                                   const double* __restrict__ alpha, const double* __restrict__ W, long ldw, long strideW,
                                                   double* __restrict__ dmean, double* __restrict__ dvar, long ldg) {
  const int c = blockIdx.x, j = blockIdx.y;
  const KernHyp h = hyp[j];
  double xc[D], gm[D], gv[D];
#pragma unroll
  for (int q = 0; q < D; ++q) {
    xc[q] = Xc[(long)(c0 + c) * D + q] / h.ls[q];
    gm[q] = 0.0;
    gv[q] = 0.0;
  }
  const double* __restrict__ X = Xs + (long)j * strideXs;
  const double* __restrict__ al = alpha + (long)j * Np;
  const double* __restrict__ Wj = W + (long)j * strideW;
  for (int i = threadIdx.x; i < N; i += 256) {
    double df[D];
    double r2 = 0.0;
#pragma unroll
    for (int q = 0; q < D; ++q) {
      df[q] = xc[q] - X[(long)i * D + q];
      r2 += df[q] * df[q];
    }
    const double f = kern_dfac(kernel_id, h.variance, r2);
    const double cm = al[i] * f, cv = -2.0 * Wj[(long)i * ldw + c] * f;
#pragma unroll
    for (int q = 0; q < D; ++q) {
      gm[q] += cm * df[q];
      gv[q] += cv * df[q];
    }
  }
  __shared__ double red[4][2 * D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < D; ++q) {
    double a = gm[q], b = gv[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      b += __shfl_xor(b, o, 64);
    }
    if (lane == 0) {
      red[w][q] = a;
      red[w][D + q] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < D) {
    const int q = threadIdx.x;
    const double il = 1.0 / h.ls[q];        // df was in scaled coordinates: (x_q - X_iq)/l_q^2 = df_q / l_q
    dmean[((long)j * ldg + c0 + c) * D + q] = (((red[0][q] + red[1][q]) + red[2][q]) + red[3][q]) * il;
    dvar[((long)j * ldg + c0 + c) * D + q] = (((red[0][D + q] + red[1][D + q]) + red[2][D + q]) + red[3][D + q]) * il;
  }
}

void launch_grad_kernel(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* Xc, int c0,
                        int Cn, const double* alpha, const double* W, long ldw, long strideW, double* dmean, double* dvar, long ldg,
                        int m, hipStream_t s, const int* kids) {
  if (Cn == 0) return;
  if (kids) {
    bocf_family_runs(kernel_id, kids, m, [&](int j0, int mr, int kid_) {
      launch_grad_kernel(Xs + (long)j0 * strideXs, strideXs, N, Np, d, kid_, hyp + j0, Xc, c0, Cn, alpha + (long)j0 * Np, W + (long)j0 * strideW, ldw, strideW,
                         dmean + (long)j0 * ldg * d, dvar + (long)j0 * ldg * d, ldg, mr, s, nullptr);
    });
    return;
  }
  dim3 grid((unsigned)Cn, (unsigned)m);
#define CASE(D)                                                                                                                   \
  case D:                                                                                                                         \
    BOCF_LAUNCH(grad_kernel<D>, grid, dim3(256), 0, s, Xs, strideXs, N, Np, kernel_id, hyp, Xc, c0, alpha, W, ldw, strideW, \
                       dmean, dvar, ldg);                                                                                         \
    break;
  switch (d) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
    CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29)
    CASE(30) CASE(31) CASE(32)
    default: break;
  }
#undef CASE
}

// ---------------------------------------------------------------------------------------------
// Small-batch path (n <= 16 candidates): the N^2 n contraction is bandwidth-bound on R, so it is done
// GEMV-style -- R is streamed once for all n right-hand sides.  NC = number of right-hand-side slots
// (1, 2, 4, 8 or 16 >= n) so a single-point call does not pay for 16.
template <int NC>
__global__ __launch_bounds__(256) void gemv_small_t_kernel(const double* __restrict__ R, long strideR, int Np,
                                                           const double* __restrict__ Kstar, long ldk, long strideK,
                                                           double* __restrict__ V) {
  // V[r][c] = sum_{kk <= r} R[kk][r] K*[kk][c]: 32 columns of R per workgroup, 8 row groups
  const int j = blockIdx.y;
  const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + l;
  const int kmax = blockIdx.x * 32 + 32;            // R is upper triangular: rows below the stripe are zero
  const double* __restrict__ Rj = R + (long)j * strideR;
  const double* __restrict__ Kj = Kstar + (long)j * strideK;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
#pragma unroll 4
  for (int kk = g; kk < kmax; kk += 8) {
    const double r = Rj[(long)kk * Np + col];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] += r * Kj[(long)kk * ldk + c];
  }
  __shared__ double red[8][32][NC + 1];
#pragma unroll
  for (int c = 0; c < NC; ++c) red[g][l][c] = acc[c];
  __syncthreads();
  for (int o = threadIdx.x; o < 32 * NC; o += 256) {
    const int cc = o % NC, ll = o / NC;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][ll][cc];
    V[((long)j * Np + blockIdx.x * 32 + ll) * NC + cc] = s;
  }
}

template <int NC>
__global__ __launch_bounds__(256) void sumsq_small_kernel(const double* __restrict__ V, int Np, double* __restrict__ sumsq, long ldo) {
  // 16 row groups whatever NC is, so a candidate's sum never depends on how many share the launch
  const int j = blockIdx.x;
  const int c = threadIdx.x % NC, g = threadIdx.x / NC;
  double acc = 0.0;
  if (g < 16)
    for (int r = g; r < Np; r += 16) {
      const double v = V[((long)j * Np + r) * NC + c];
      acc += v * v;
    }
  __shared__ double red[16][16];
  if (g < 16) red[g][c] = acc;
  __syncthreads();
  if (threadIdx.x < NC) {
    double s = 0.0;
    for (int q = 0; q < 16; ++q) s += red[q][threadIdx.x];
    sumsq[(long)j * ldo + threadIdx.x] = s;
  }
}

// W[r][c] = sum_{kk >= r} R[r][kk] V[kk][c]: one wave per row, lanes stride the columns of R
template <int NC>
__global__ __launch_bounds__(256) void gemv_small_n_kernel(const double* __restrict__ R, long strideR, int Np,
                                                           const double* __restrict__ V, double* __restrict__ W) {
  const int j = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + w;
  const double* __restrict__ Rr = R + (long)j * strideR + (long)r * Np;
  const double* __restrict__ Vj = V + (long)j * Np * NC;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
#pragma unroll 4
  for (int kk = (r & ~63) + lane; kk < Np; kk += 64) {
    const double rv = Rr[kk];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] += rv * Vj[(long)kk * NC + c];
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) W[((long)j * Np + r) * NC + c] = acc[c];
  }
}

// ---------------------------------------------------------------------------------------------
// The same small path on the matrix pipe (round 3): what the multi-start refinement calls hundreds of times per BO iteration with
// 1 ... 16 points.  Measured before (N = 1024, m = 4, 16 points, rocprofv3): cross kernel 45 us (its column-parallel map leaves eight
// lanes of one wave per workgroup with work), V = R^T K* 41 us (16 broadcast loads of K* per row and thread), W = R V 62 us (every wave
// re-reads all of V: 270 MB of L2 traffic), column sums of squares 16.5 us in a launch of their own.
//
// K* and the posterior-mean partials of <= 16 candidates: a workgroup owns one 128-row block of one output; thread (row, half) computes the
// kernel values of eight candidates for its row, then lane c adds the block's 128 products of candidate c up in row order with the
// same compensated accumulation as cross_kernel -- the same partial per (block, candidate), bit for bit, so a point's mean does not
// depend on which path evaluated it.
template <int D, int KID>
__global__ __launch_bounds__(256) void cross_small_kernel(const double* __restrict__ Xs, long strideXs, int N, int Np, const KernHyp* __restrict__ hyp,
                                                          const double* __restrict__ Xc, int c0, int Cn, int nc, const double* __restrict__ alpha,
                                                          double* __restrict__ Kstar, long ldk, long strideK, double* __restrict__ meanpart,
                                                          double* __restrict__ meanlo, int Cpad, int jbase, int mtot) {
  const int j = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  __shared__ double kv[BOCF_TILE][17];
  __shared__ double xc[16][D];
  const KernHyp h = hyp[j];
  for (int i = tid; i < 16 * D; i += 256) {
    const int c = i / D, q = i - c * D;
    xc[c][q] = c < Cn ? Xc[(long)(c0 + c) * D + q] / h.ls[q] : 0.0;
  }
  __syncthreads();
  const int row = tid & 127, half = tid >> 7;
  const int kk = blk * BOCF_TILE + row;
  const double* __restrict__ X = Xs + (long)j * strideXs;
  double* __restrict__ Kj = Kstar + (long)j * strideK;
  double xr[D];
#pragma unroll
  for (int q = 0; q < D; ++q) xr[q] = kk < N ? X[(long)kk * D + q] : 0.0;
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const int c = half * 8 + cc;
    if (c < nc) {
      double v0 = 0.0;
      if (kk < N && c < Cn) {
        double r0 = 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) {
          const double d0 = xr[q] - xc[c][q];
          r0 += d0 * d0;
        }
        v0 = kern_of_r2_p(KID, h.variance, r0);
      }
      Kj[(long)kk * ldk + c] = v0;
      kv[row][c] = v0;
    }
  }
  __syncthreads();
  if (tid < Cn) {
    const double* __restrict__ al = alpha + (long)j * Np;
    double mean0 = 0.0, lo0 = 0.0;
    for (int r = 0; r < BOCF_TILE; ++r) {
      const int k2 = blk * BOCF_TILE + r;
      if (k2 < N) dd_fma_acc(mean0, lo0, kv[r][tid], al[k2]);
    }
    meanpart[((long)blk * mtot + jbase + j) * Cpad + tid] = mean0;
    meanlo[((long)blk * mtot + jbase + j) * Cpad + tid] = lo0;
  }
}

template <int D>
static void launch_cross_small_d(const double* Xs, long strideXs, int N, int Np, int kernel_id, const KernHyp* hyp, const double* Xc, int c0, int Cn, int nc,
                                 const double* alpha, double* Kstar, long ldk, long strideK, double* meanpart, double* meanlo, int Cpad, int m, hipStream_t s,
                                 int jbase, int mtot) {
  const dim3 grid((unsigned)(Np / BOCF_TILE), (unsigned)m);
  const int kid = kernel_id <= 1 ? 0 : kernel_id;
#define LAUNCH(KID) BOCF_LAUNCH((cross_small_kernel<D, KID>), grid, dim3(256), 0, s, Xs, strideXs, N, Np, hyp, Xc, c0, Cn, nc, alpha, Kstar, ldk, strideK, \
                                meanpart, meanlo, Cpad, jbase, mtot)
  if (kid == 0) LAUNCH(0);
  else if (kid == 2) LAUNCH(2);
  else LAUNCH(3);
#undef LAUNCH
}

void launch_cross_small(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* Xc, int c0, int Cn, int nc,
                        const double* alpha, double* Kstar, long ldk, long strideK, double* meanpart, double* meanlo, int Cpad, int m, hipStream_t s,
                        const int* kids) {
  bocf_family_runs(kernel_id, kids, m, [&](int j0, int mr, int kid_) {
#define CASE(D)                                                                                                                                  \
  case D:                                                                                                                                        \
    launch_cross_small_d<D>(Xs + (long)j0 * strideXs, strideXs, N, Np, kid_, hyp + j0, Xc, c0, Cn, nc, alpha + (long)j0 * Np, Kstar + (long)j0 * strideK, ldk, \
                            strideK, meanpart, meanlo, Cpad, mr, s, j0, m);                                                                        \
    break;
    switch (d) {
      CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
      CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29) CASE(30) CASE(31)
      CASE(32)
      default: break;
    }
#undef CASE
  });
}

// V = R^T K* (NC <= 16 columns) as a skinny MFMA product: a wave owns 16 rows of V, A operand R[kk][r0 + m] (the k-major rows of the
// upper factor's inverse: 128 contiguous bytes per 16 lanes), B operand K*[kk][n] (zero beyond NC), kk = 0 ... r0 + 15 (R is upper
// triangular with explicit zeros).  The wave also leaves the sum of squares of its 16 rows per column in sumsq[j][tile][c]:
// finalize_var adds the Np / 16 partials in tile order, so a candidate's variance does not depend on how many share the launch.
template <int NC>
__global__ __launch_bounds__(512) void gemv_small_t_mfma_kernel(const double* __restrict__ R, long strideR, int Np, const double* __restrict__ Kstar,
                                                                long ldk, long strideK, double* __restrict__ V, double* __restrict__ sumsq,
                                                                long ldss) {
  // one workgroup per 16-row tile, its eight waves take every eighth k4-step (the contraction is a chain of dependent loads per wave: eight
  // shorter chains, 16 steps in flight each), partial tiles added in wave order through LDS
  typedef double v4d_p __attribute__((ext_vector_type(4)));
  __shared__ double part[8][16][17];
  const int j = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c15 = lane & 15, q4 = lane >> 4;
  const int tile = blockIdx.x;
  const int r0 = tile * 16;
  const double* __restrict__ Rj = R + (long)j * strideR + r0 + c15;
  const double* __restrict__ Kj = Kstar + (long)j * strideK + (c15 < NC ? c15 : 0);
  v4d_p acc = (v4d_p){0.0, 0.0, 0.0, 0.0};
  const int nk4 = (r0 + 16) >> 2;
#pragma unroll 16
  for (int k4 = w; k4 < nk4; k4 += 8) {
    const long kk = 4 * k4 + q4;
    const double a = Rj[kk * Np];
    const double b = c15 < NC ? Kj[kk * ldk] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[w][4 * r + q4][c15] = acc[r];
  __syncthreads();
  if (w == 0) {
    double ss = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * r + q4;
      double v = part[0][row][c15];
#pragma unroll
      for (int q = 1; q < 8; ++q) v += part[q][row][c15];
      if (c15 < NC) V[((long)j * Np + r0 + row) * NC + c15] = v;
      ss = __builtin_fma(v, v, ss);
    }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (q4 == 0 && c15 < NC) sumsq[((long)j * (Np / 16) + tile) * ldss + c15] = ss;
  }
}

// W = R V (NC <= 16 columns): A operand from R^T (RT[kk][r0 + m] = R[r0 + m][kk]: contiguous again), kk = r0 ... Np - 1; same split.
template <int NC>
__global__ __launch_bounds__(512) void gemv_small_n_mfma_kernel(const double* __restrict__ RT, long strideR, int Np, const double* __restrict__ V,
                                                                double* __restrict__ W) {
  typedef double v4d_p __attribute__((ext_vector_type(4)));
  __shared__ double part[8][16][17];
  const int j = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c15 = lane & 15, q4 = lane >> 4;
  const int r0 = blockIdx.x * 16;
  const double* __restrict__ Tj = RT + (long)j * strideR + r0 + c15;
  const double* __restrict__ Vj = V + (long)j * Np * NC + (c15 < NC ? c15 : 0);
  v4d_p acc = (v4d_p){0.0, 0.0, 0.0, 0.0};
#pragma unroll 16
  for (int k4 = (r0 >> 2) + w; k4 < (Np >> 2); k4 += 8) {
    const long kk = 4 * k4 + q4;
    const double a = Tj[kk * Np];
    const double b = c15 < NC ? Vj[kk * NC] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[w][4 * r + q4][c15] = acc[r];
  __syncthreads();
  if (w == 0 && c15 < NC) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * r + q4;
      double v = part[0][row][c15];
#pragma unroll
      for (int q = 1; q < 8; ++q) v += part[q][row][c15];
      W[((long)j * Np + r0 + row) * NC + c15] = v;
    }
  }
}

#define SMALL_DISPATCH(nc, CALL) \
  switch (nc) {                  \
    case 1: { constexpr int NC = 1; CALL; } break;   \
    case 2: { constexpr int NC = 2; CALL; } break;   \
    case 4: { constexpr int NC = 4; CALL; } break;   \
    case 8: { constexpr int NC = 8; CALL; } break;   \
    default: { constexpr int NC = 16; CALL; } break; \
  }

void launch_gemv_small_t(const double* R, long strideR, int Np, const double* Kstar, long ldk, long strideK, double* V, int nc, int m,
                         hipStream_t s) {
  SMALL_DISPATCH(nc, BOCF_LAUNCH(gemv_small_t_kernel<NC>, dim3((unsigned)(Np / 32), (unsigned)m), dim3(256), 0, s, R, strideR, Np,
                                        Kstar, ldk, strideK, V))
}

void launch_gemv_small_t_mfma(const double* R, long strideR, int Np, const double* Kstar, long ldk, long strideK, double* V, double* sumsq, long ldss,
                              int nc, int m, hipStream_t s) {
  SMALL_DISPATCH(nc, BOCF_LAUNCH(gemv_small_t_mfma_kernel<NC>, dim3((unsigned)(Np / 16), (unsigned)m), dim3(512), 0, s, R, strideR, Np, Kstar,
                                 ldk, strideK, V, sumsq, ldss))
}

void launch_gemv_small_n_mfma(const double* RT, long strideR, int Np, const double* V, double* W, int nc, int m, hipStream_t s) {
  SMALL_DISPATCH(nc, BOCF_LAUNCH(gemv_small_n_mfma_kernel<NC>, dim3((unsigned)(Np / 16), (unsigned)m), dim3(512), 0, s, RT, strideR, Np, V, W))
}

void launch_sumsq_small(const double* V, int Np, double* sumsq, long ldo, int nc, int m, hipStream_t s) {
  SMALL_DISPATCH(nc, BOCF_LAUNCH(sumsq_small_kernel<NC>, dim3((unsigned)m), dim3(256), 0, s, V, Np, sumsq, ldo))
}

void launch_gemv_small_n(const double* R, long strideR, int Np, const double* V, double* W, int nc, int m, hipStream_t s) {
  SMALL_DISPATCH(nc, BOCF_LAUNCH(gemv_small_n_kernel<NC>, dim3((unsigned)(Np / 4), (unsigned)m), dim3(256), 0, s, R, strideR, Np, V, W))
}
