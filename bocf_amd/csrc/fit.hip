// Fit-side kernels around the factorization: input scaling, K(X,X) build, block copies / transposes / tile packing, the GEMVs of
// alpha, its double-double refinement, log-marginal, hyper-gradient sums, rank-1 append.  (The factorization chain kernels are potrf.hip and
// chol_team.hip, the fused small-model inference / HMC chain infer128.hip, the stream-resident chain hmc_stream.hip.)
// Storage convention: every N x N matrix is padded to Np (multiple of 128) and held ROW-MAJOR in
// its UPPER form: Ky = U^T U with U upper triangular (U = L^T of the reference's lower factor,
// GPy/util/linalg.py:52-55), R = U^-1 upper.  Padding rows/cols carry the identity.
#include "bocf_internal.h"
#include "fit_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

// ---------------------------------------------------------------------------------------------
// Xs[j][i][q] = X[i][q] / l_jq   (ARD scaling of the inputs, stationary.py:161-164 / se.py:88-91)
__global__ void scale_inputs_kernel(const double* __restrict__ X, int n, int d, const KernHyp* __restrict__ hyp,
                                    double* __restrict__ Xs, long strideXs) {
  const int j = blockIdx.y;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * d) return;
  const int q = (int)(idx % d);
  Xs[(long)j * strideXs + idx] = X[idx] / hyp[j].ls[q];
}

void launch_scale_inputs(const double* X, int n, int d, const KernHyp* hyp, int m, double* Xs, long strideXs, hipStream_t s) {
  if (n == 0) return;
  dim3 grid((unsigned)(((long)n * d + 255) / 256), (unsigned)m);
  BOCF_LAUNCH(scale_inputs_kernel, grid, dim3(256), 0, s, X, n, d, hyp, Xs, strideXs);
}

// ---------------------------------------------------------------------------------------------
// K(X,X): one 64 (rows) x 512 (columns) tile per workgroup; every thread owns TWO adjacent column points (coordinates in
// registers), the row point is wave-uniform (scalar loads), so the inner loop is 2 d subtract + 2 d fma + two independent
// exp chains and one 16-B-per-lane store of a contiguous 4 KiB row segment per workgroup and row: HBM-write bound
// (8 B per element; only tiles on/above the diagonal are produced).  Per element the arithmetic is that of a one-column thread.
typedef double v2d_f __attribute__((ext_vector_type(2)));
#ifndef BT_ROWS
#define BT_ROWS 64           // rows of K per workgroup (a multiple of 2 that divides 128)
#endif
template <int D, int KID>
__global__ __launch_bounds__(256) void build_train_lds_kernel(const double* __restrict__ Xs, long strideXs, int N, int Np,
                                                              const KernHyp* __restrict__ hyp, const double* __restrict__ jitter, int add_diag,
                                                              double* __restrict__ S, long strideS) {
  __shared__ double xr[BT_ROWS * D];
  // one workgroup per tile that touches the diagonal or lies above it (no workgroups for the strictly lower tiles: a third of the square grid's
  // workgroups used to exit at once): tiles are numbered row by row, row block rb holding column blocks rb * BT_ROWS / 512 ... ncb - 1
  const int j = blockIdx.z;
  const int ncb = (Np + 511) / 512;
  int bx, by;
  {
    constexpr int RPC = 512 / BT_ROWS;                     // row blocks that start in the same column block
    int u = blockIdx.x, gq = 0;
    while (u >= RPC * (ncb - gq)) {                        // group gq = RPC row blocks of ncb - gq tiles each
      u -= RPC * (ncb - gq);
      ++gq;
    }
    const int rin = u / (ncb - gq);
    by = gq * RPC + rin;
    bx = gq + (u - rin * (ncb - gq));
  }
  const int r0 = by * BT_ROWS;
  const int gc = (bx * 256 + threadIdx.x) * 2;
  const double* __restrict__ X = Xs + (long)j * strideXs;
  for (int idx = threadIdx.x; idx < BT_ROWS * D; idx += 256) {
    const int r = r0 + idx / D;
    xr[idx] = r < N ? X[(long)r0 * D + idx] : 0.0;
  }
  const double variance = hyp[j].variance;
  const double dg = add_diag ? (hyp[j].noise + 1e-8 + (jitter ? jitter[j] : 0.0)) : 0.0;
  double xa[D], xb[D];
#pragma unroll
  for (int q = 0; q < D; ++q) {
    xa[q] = gc < N ? X[(long)gc * D + q] : 0.0;
    xb[q] = gc + 1 < N ? X[(long)(gc + 1) * D + q] : 0.0;
  }
  __syncthreads();
  if (gc >= Np) return;
  double* __restrict__ Sj = S + (long)j * strideS;
  const bool c0 = gc < N, c1 = gc + 1 < N;
  // interior tile (uniform per workgroup): every row and column is a real point and the diagonal does not cross it -- no padding /
  // diagonal selects (7 of the ~50 vector instructions per element; same values)
  if (r0 + BT_ROWS <= N && (bx + 1) * 512 <= N && bx * 512 > r0 + BT_ROWS - 1) {
#pragma unroll 1
    for (int rr = 0; rr < BT_ROWS; rr += 4) {
      double v[4][2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double ra = 0.0, rb = 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) {
          const double xq = xr[(rr + e) * D + q];
          const double d0 = xq - xa[q], d1 = xq - xb[q];
          ra += d0 * d0;
          rb += d1 * d1;
        }
        v[e][0] = kern_of_r2(KID, variance, ra);
        v[e][1] = kern_of_r2(KID, variance, rb);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) *reinterpret_cast<v2d_f*>(Sj + (long)(r0 + rr + e) * Np + gc) = (v2d_f){v[e][0], v[e][1]};
    }
    return;
  }
#pragma unroll 1
  for (int rr = 0; rr < BT_ROWS; rr += 4) {
    double v[4][2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int gr = r0 + rr + e;
      double ra = 0.0, rb = 0.0;
#pragma unroll
      for (int q = 0; q < D; ++q) {
        const double xq = xr[(rr + e) * D + q];
        const double d0 = xq - xa[q], d1 = xq - xb[q];
        ra += d0 * d0;
        rb += d1 * d1;
      }
      const double k0 = kern_of_r2(KID, variance, ra), k1 = kern_of_r2(KID, variance, rb);
      const bool rreal = gr < N;
      v[e][0] = rreal ? (gr == gc ? variance + dg : (c0 ? k0 : 0.0)) : (gr == gc ? 1.0 : 0.0);       // identity padding
      v[e][1] = rreal ? (gr == gc + 1 ? variance + dg : (c1 ? k1 : 0.0)) : (gr == gc + 1 ? 1.0 : 0.0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<v2d_f*>(Sj + (long)(r0 + rr + e) * Np + gc) = (v2d_f){v[e][0], v[e][1]};
  }
}

void launch_build_train_kernel(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp,
                               const double* jitter, int add_diag, double* S, long strideS, int m, hipStream_t s, const int* kids) {
  if (kids) {
    bocf_family_runs(kernel_id, kids, m, [&](int j0, int mr, int kid_) {
      launch_build_train_kernel(Xs + (long)j0 * strideXs, strideXs, N, Np, d, kid_, hyp + j0, jitter ? jitter + j0 : nullptr, add_diag, S + (long)j0 * strideS,
                                strideS, mr, s, nullptr);
    });
    return;
  }
  // tiles on / above the diagonal: row block rb (BT_ROWS rows) holds the 512-column blocks from rb * BT_ROWS / 512 on
  unsigned ntiles = 0;
  for (int rb = 0; rb < Np / BT_ROWS; ++rb) ntiles += (unsigned)((Np + 511) / 512 - rb * BT_ROWS / 512);
  dim3 grid(ntiles, 1, (unsigned)m);
  const int kid = kernel_id <= 1 ? 0 : kernel_id;
#define LAUNCH(D, KID) BOCF_LAUNCH((build_train_lds_kernel<D, KID>), grid, dim3(256), 0, s, Xs, strideXs, N, Np, hyp, jitter, add_diag, S, strideS)
#define CASE(D)                       \
  case D:                             \
    if (kid == 0) LAUNCH(D, 0);       \
    else if (kid == 2) LAUNCH(D, 2);  \
    else LAUNCH(D, 3);                \
    break;
  switch (d) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
    CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29)
    CASE(30) CASE(31) CASE(32)
    default: break;
  }
#undef CASE
#undef LAUNCH
}

// ---------------------------------------------------------------------------------------------
__global__ void copy_diag_blocks_kernel(const double* __restrict__ E, long strideE, double* __restrict__ R, long strideR, int Np, int blk_lo) {
  const int j = blockIdx.z, p = blk_lo + blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // < 128*128
  const int r = i >> 7, c = i & 127;
  R[(long)j * strideR + (long)(p * NB + r) * Np + p * NB + c] = E[(long)j * strideE + (long)p * NB * NB + i];
}

void launch_copy_diag_blocks(const double* E, long strideE, double* R, long strideR, int Np, int blk_lo, int blk_hi, int m, hipStream_t s) {
  if (blk_hi <= blk_lo) return;
  BOCF_LAUNCH(copy_diag_blocks_kernel, dim3(NB * NB / 256, (unsigned)(blk_hi - blk_lo), (unsigned)m), dim3(256), 0, s, E, strideE, R, strideR, Np,
              blk_lo);
}

// dst[c][r] = src[r][c] over `count` rows x cols blocks whose corners advance by `step` along the diagonal
__global__ __launch_bounds__(256) void transpose_block_kernel(const double* __restrict__ src, double* __restrict__ dst, long stride, int Np,
                                                              int r0, int c0, int rows, int cols, int count, int step) {
  __shared__ double t[32][33];
  const int tiles_c = cols / 32;
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
  const int blk = blockIdx.y % count, j = blockIdx.y / count;
  const long rb = r0 + (long)blk * step + tr * 32, cb = c0 + (long)blk * step + tc * 32;
  const double* __restrict__ sj = src + (long)j * stride;
  double* __restrict__ dj = dst + (long)j * stride;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  for (int r = ly; r < 32; r += 8) t[r][lx] = sj[(rb + r) * Np + cb + lx];
  __syncthreads();
  for (int r = ly; r < 32; r += 8) dj[(cb + r) * Np + rb + lx] = t[lx][r];
}

void launch_transpose_block(const double* src, double* dst, long stride, int Np, int r0, int c0, int rows, int cols, int count, int step,
                            int m, hipStream_t s) {
  if (rows <= 0 || cols <= 0 || count <= 0) return;
  dim3 grid((unsigned)((rows / 32) * (cols / 32)), (unsigned)(count * m));
  BOCF_LAUNCH(transpose_block_kernel, grid, dim3(256), 0, s, src, dst, stride, Np, r0, c0, rows, cols, count, step);
}

// ---------------------------------------------------------------------------------------------
// Iterative refinement of alpha (declared in bocf_internal.h).  kalpha_dd: one thread per row i of Ky, looping over one 128-block of
// columns k (wave-uniform scalar loads of X_k and alpha_k), the sum carried as a pair; Ky[i][k] is rebuilt with the arithmetic of
// build_train_kernel, so the residual is that of the matrix the factorization actually saw.
template <int D, int KID>
__global__ __launch_bounds__(256) void kalpha_dd_kernel(const double* __restrict__ Xs, long strideXs, int N, int Np, const KernHyp* __restrict__ hyp,
                                                        const double* __restrict__ jitter, const double* __restrict__ alpha, double* __restrict__ part, int kb) {
  const int j = blockIdx.z, blk = blockIdx.y, nblk = gridDim.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Np) return;
  const double* __restrict__ X = Xs + (long)j * strideXs;
  const double* __restrict__ al = alpha + (long)j * Np;
  const double variance = hyp[j].variance;
  const double dg = hyp[j].noise + 1e-8 + (jitter ? jitter[j] : 0.0);
  double xi[D];
#pragma unroll
  for (int q = 0; q < D; ++q) xi[q] = i < N ? X[(long)i * D + q] : 0.0;
  double s = 0.0, c = 0.0;
  if (i < N) {
    const int kend = (blk + 1) * kb < N ? (blk + 1) * kb : N;
    for (int k = blk * kb; k < kend; ++k) {
      double r2 = 0.0;
#pragma unroll
      for (int q = 0; q < D; ++q) {
        const double t = X[(long)k * D + q] - xi[q];
        r2 += t * t;
      }
      const double kv = k == i ? variance + dg : kern_of_r2(KID, variance, r2);
      dd_fma_acc(s, c, kv, al[k]);
    }
  }
  double* o = part + (((long)j * nblk + blk) * 2) * Np + i;
  o[0] = s;
  o[Np] = c;
}

// columns per partial sum: 128 from 4096 rows (thousands of workgroups either way), 64 from 2048, 32 below -- at N = 1024 the 128 workgroups of 128-column
// blocks leave the chip half empty and every thread runs 128 dependent pair-accumulations in a row (64 us; 32-column blocks: 4x the workgroups)
int kalpha_block(int Np) { return Np >= 4096 ? NB : (Np >= 2048 ? 64 : 32); }
void launch_kalpha_dd(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* jitter,
                      const double* alpha, double* part, int m, hipStream_t s, const int* kids) {
  const int kb = kalpha_block(Np);
  if (kids) {
    bocf_family_runs(kernel_id, kids, m, [&](int j0, int mr, int kid_) {
      launch_kalpha_dd(Xs + (long)j0 * strideXs, strideXs, N, Np, d, kid_, hyp + j0, jitter ? jitter + j0 : nullptr, alpha + (long)j0 * Np,
                       part + (long)j0 * (Np / kb) * 2 * Np, mr, s, nullptr);
    });
    return;
  }
  dim3 grid((unsigned)((Np + 255) / 256), (unsigned)(Np / kb), (unsigned)m);
  const int kid = kernel_id <= 1 ? 0 : kernel_id;
#define LAUNCH(D, KID) BOCF_LAUNCH((kalpha_dd_kernel<D, KID>), grid, dim3(256), 0, s, Xs, strideXs, N, Np, hyp, jitter, alpha, part, kb)
#define CASE(D)                       \
  case D:                             \
    if (kid == 0) LAUNCH(D, 0);       \
    else if (kid == 2) LAUNCH(D, 2);  \
    else LAUNCH(D, 3);                \
    break;
  switch (d) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
    CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29)
    CASE(30) CASE(31) CASE(32)
    default: break;
  }
#undef CASE
#undef LAUNCH
}

__global__ void refine_rhs_kernel(const double* __restrict__ part, int N, int Np, const double* __restrict__ yc, double* __restrict__ r, int nblk) {
  const int j = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Np) return;
  double s = 0.0, c = 0.0;
  for (int blk = 0; blk < nblk; ++blk) {
    const double* o = part + (((long)j * nblk + blk) * 2) * Np + i;
    dd_add_acc(s, c, o[0], o[Np]);
  }
  r[(long)j * Np + i] = i < N ? (yc[(long)j * Np + i] - s) - c : 0.0;
}
void launch_refine_rhs(const double* part, int N, int Np, const double* yc, double* r, int m, hipStream_t s) {
  BOCF_LAUNCH(refine_rhs_kernel, dim3((unsigned)((Np + 255) / 256), (unsigned)m), dim3(256), 0, s, part, N, Np, yc, r, Np / kalpha_block(Np));
}

__global__ void refine_apply_kernel(const double* __restrict__ delta, int N, int Np, const KernHyp* __restrict__ hyp, const double* __restrict__ jitter,
                                    const double* __restrict__ yc, double* __restrict__ alpha, double* __restrict__ mu_train, long ldmu) {
  const int j = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double a = alpha[(long)j * Np + i] + (delta ? delta[(long)j * Np + i] : 0.0);
  alpha[(long)j * Np + i] = a;
  if (mu_train) {
    // what the fit added to the diagonal, as the stored matrix holds it: fl(variance + dg) - variance
    const double variance = hyp[j].variance;
    const double dg = (variance + (hyp[j].noise + 1e-8 + (jitter ? jitter[j] : 0.0))) - variance;
    mu_train[(long)j * ldmu + i] = __builtin_fma(-dg, a, yc[(long)j * Np + i]) + hyp[j].ymean;
  }
}
void launch_refine_apply(const double* delta, int N, int Np, const KernHyp* hyp, const double* jitter, const double* yc, double* alpha,
                         double* mu_train, long ldmu, int m, hipStream_t s) {
  BOCF_LAUNCH(refine_apply_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)m), dim3(256), 0, s, delta, N, Np, hyp, jitter, yc, alpha, mu_train,
              ldmu);
}

// Upper 128 x 128 tiles of one Np x Np matrix <-> a contiguous buffer (tile (r, c), c >= r, at index r nb - r (r - 1) / 2 + (c - r)):
// what an output-sharded fit sends per inverse factor ("L_j lower-tri", 67 MB instead of 134 at N = 4096).  One
// workgroup per tile, 16-B accesses, rows of a tile contiguous on both sides.
template <int UNPACK>
__global__ __launch_bounds__(256) void upper_tiles_kernel(const double* __restrict__ src, double* __restrict__ dst, int Np) {
  const int nb = Np / NB;
  int r = 0, t = blockIdx.x;
  while (t >= nb - r) {            // (nb <= a few hundred: a scalar loop)
    t -= nb - r;
    ++r;
  }
  const int c = r + t;
  const long full = (long)r * NB * Np + (long)c * NB, pk = (long)blockIdx.x * NB * NB;
  typedef double v2 __attribute__((ext_vector_type(2)));
  for (int e = threadIdx.x; e < NB * NB / 2; e += 256) {
    const int row = e / (NB / 2), col = (e % (NB / 2)) * 2;
    if (UNPACK) *reinterpret_cast<v2*>(dst + full + (long)row * Np + col) = *reinterpret_cast<const v2*>(src + pk + row * NB + col);
    else *reinterpret_cast<v2*>(dst + pk + row * NB + col) = *reinterpret_cast<const v2*>(src + full + (long)row * Np + col);
  }
}
void launch_pack_upper_tiles(const double* R, int Np, double* packed, hipStream_t s) {
  const int nb = Np / NB;
  BOCF_LAUNCH(upper_tiles_kernel<0>, dim3((unsigned)(nb * (nb + 1) / 2)), dim3(256), 0, s, R, packed, Np);
}
void launch_unpack_upper_tiles(const double* packed, int Np, double* R, hipStream_t s) {
  const int nb = Np / NB;
  BOCF_LAUNCH(upper_tiles_kernel<1>, dim3((unsigned)(nb * (nb + 1) / 2)), dim3(256), 0, s, packed, R, Np);
}

// alpha[r] = sum_{kk >= r} R[r][kk] t[kk]   (one wave per row)
__global__ __launch_bounds__(256) void gemv_upper_n_kernel(const double* __restrict__ R, long strideR, int Np,
                                                           const double* __restrict__ t, double* __restrict__ alpha) {
  const int j = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + w;
  const double* __restrict__ Rr = R + (long)j * strideR + (long)r * Np;
  const double* __restrict__ tj = t + (long)j * Np;
  double acc = 0.0;
  for (int kk = (r & ~63) + lane; kk < Np; kk += 64) acc += Rr[kk] * tj[kk];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) alpha[(long)j * Np + r] = acc;
}

void launch_gemv_upper_n(const double* R, long strideR, int Np, const double* t, double* alpha, int m, hipStream_t s) {
  BOCF_LAUNCH(gemv_upper_n_kernel, dim3((unsigned)(Np / 4), (unsigned)m), dim3(256), 0, s, R, strideR, Np, t, alpha);
}

// ---------------------------------------------------------------------------------------------
// log-marginal 0.5 (-N log 2pi - logdet - alpha.y), logdet = 2 sum log U_ii
// (exact_gaussian_inference.py:53, linalg.py:202)
__global__ __launch_bounds__(256) void lml_kernel(const double* __restrict__ S, long strideS, int N, int Np,
                                                  const double* __restrict__ alpha, const double* __restrict__ yc,
                                                  double* __restrict__ lml) {
  const int j = blockIdx.x;
  double ld = 0.0, dt = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) {
    ld += log(S[(long)j * strideS + (long)i * Np + i]);
    dt += alpha[(long)j * Np + i] * yc[(long)j * Np + i];
  }
  __shared__ double r1[256], r2[256];
  r1[threadIdx.x] = ld;
  r2[threadIdx.x] = dt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      r1[threadIdx.x] += r1[threadIdx.x + o];
      r2[threadIdx.x] += r2[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) lml[j] = 0.5 * (-(double)N * 1.8378770664093454836 - 2.0 * r1[0] - r2[0]);
}

void launch_lml(const double* S, long strideS, int N, int Np, const double* alpha, const double* yc, double* lml, int m, hipStream_t s) {
  BOCF_LAUNCH(lml_kernel, dim3((unsigned)m), dim3(256), 0, s, S, strideS, N, Np, alpha, yc, lml);
}

// ---------------------------------------------------------------------------------------------
// Gradients of the log marginal likelihood (numerical core):
//   dL_dK = 0.5 (alpha alpha^T - Ky^-1)                        exact_gaussian_inference.py:61
//   d/dnoise    = sum_i dL_dK_ii                                :63, gaussian.py:71-72
//   d/dvariance = sum_ij dL_dK_ij K_ij / variance               stationary.py:197, se.py:181
//   d/dl_q      = sum_ij dL_dK_ij f_ij (x_iq - x_jq)^2 / l_q^3  stationary.py:203-212,236-237, se.py:183
// with f = -invdist dK_dr (RBF/SE: k;  Matern52: (5/3) s2 (1 + sqrt5 r) e^{-sqrt5 r};  Matern32: 3 s2 e^{-sqrt3 r}).
// Ky^-1 = R R^T comes from a triangular GEMM (upper tiles); every (i<j) pair is visited once and counted twice.
// One 64 x 256 tile per workgroup, thread per column; per-workgroup partials are reduced in a fixed order.
__device__ __forceinline__ double kern_hfac(int kid, double variance, double r2) {
  if (kid <= 1) return variance * bocf_exp_nonpos(-0.5 * r2);
  const double r = sqrt(r2);
  if (kid == 2) {
    const double s5r = 2.23606797749978969641 * r;
    return (5.0 / 3.0) * variance * (1.0 + s5r) * bocf_exp_nonpos(-s5r);
  }
  return 3.0 * variance * bocf_exp_nonpos(-1.73205080756887729353 * r);
}

// rows per workgroup: 16 up to 1024 points (a thread walks its rows one dependent exp after the other: 64 rows are 34 us at N = 256 whatever the
// size of the grid), 64 above
static inline int hypgrad_rows(int Np) { return Np <= 1024 ? 16 : 64; }
int hypgrad_num_blocks(int Np) { return ((Np + 255) / 256) * (Np / hypgrad_rows(Np)); }

template <int D, int KID>
__global__ __launch_bounds__(256) void hypgrad_kernel(const double* __restrict__ Xs, long strideXs, int N, int Np,
                                                      const KernHyp* __restrict__ hyp, const double* __restrict__ alpha,
                                                      const double* __restrict__ Kinv, long strideK, double* __restrict__ part, int rows) {
  const int j = blockIdx.z;
  const int r0 = blockIdx.y * rows;
  const int gc = blockIdx.x * 256 + threadIdx.x;
  const int nblk = gridDim.x * gridDim.y;
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  double sv = 0.0, sn = 0.0, sl[D];
#pragma unroll
  for (int q = 0; q < D; ++q) sl[q] = 0.0;
  const bool active = !(blockIdx.x * 256 + 255 < r0) && gc < N;
  if (active) {
    const double* __restrict__ X = Xs + (long)j * strideXs;
    const double* __restrict__ al = alpha + (long)j * Np;
    const double* __restrict__ Kj = Kinv + (long)j * strideK;
    const double variance = hyp[j].variance;
    double xc[D];
#pragma unroll
    for (int q = 0; q < D; ++q) xc[q] = X[(long)gc * D + q];
    const double ac = al[gc];
    for (int rr = 0; rr < rows; ++rr) {
      const int gr = r0 + rr;
      if (gr >= N || gr > gc) break;                     // upper triangle only (rows ascend)
      const double g = 0.5 * (al[gr] * ac - Kj[(long)gr * Np + gc]);
      if (gr == gc) {
        sn += g;
        sv += g;                                         // K_ii / variance = 1
      } else {
        double df[D];
        double r2 = 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) {
          df[q] = X[(long)gr * D + q] - xc[q];
          r2 += df[q] * df[q];
        }
        double kv, f;
        if (KID <= 1) {
          kv = variance * bocf_exp_nonpos(-0.5 * r2);
          f = kv;
        } else {
          const double r = sqrt(r2);
          if (KID == 2) {
            const double s5r = 2.23606797749978969641 * r, e = bocf_exp_nonpos(-s5r);
            kv = variance * (1.0 + s5r + (5.0 / 3.0) * r2) * e;
            f = (5.0 / 3.0) * variance * (1.0 + s5r) * e;
          } else {
            const double s3r = 1.73205080756887729353 * r, e = bocf_exp_nonpos(-s3r);
            kv = variance * (1.0 + s3r) * e;
            f = 3.0 * variance * e;
          }
        }
        const double g2 = 2.0 * g;                       // (i,j) and (j,i)
        sv += g2 * kv / variance;
        const double gf = g2 * f;
#pragma unroll
        for (int q = 0; q < D; ++q) sl[q] += gf * (df[q] * df[q]);
      }
    }
  }
  // block reduction in a fixed order: wave butterflies, then the 4 waves
  __shared__ double red[4][2 + D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double vals[2 + D];
  vals[0] = sv;
  vals[1] = sn;
#pragma unroll
  for (int q = 0; q < D; ++q) vals[2 + q] = sl[q];
#pragma unroll
  for (int t = 0; t < 2 + D; ++t) {
    double v = vals[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[w][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 2 + D) {
    const int t = threadIdx.x;
    part[((long)j * nblk + blk) * (2 + D) + t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
  }
}

__global__ __launch_bounds__(64) void hypgrad_reduce_kernel(const double* __restrict__ part, int nblk, int d, const KernHyp* __restrict__ hyp,
                                                            double* __restrict__ out) {
  const int j = blockIdx.x, lane = threadIdx.x;
  for (int t = 0; t < 2 + d; ++t) {
    double s = hypgrad_partial_sum(part, j, nblk, 2 + d, t, lane);
    if (t >= 2) s /= hyp[j].ls[t - 2];   // differences were in scaled coordinates: (dx/l)^2 / l = dx^2 / l^3
    if (lane == 0) out[(long)j * (2 + d) + t] = s;
  }
}

void launch_hypgrad(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* alpha,
                    const double* Kinv, long strideK, double* part, double* out, int m, hipStream_t s, const int* kids, bool reduce) {
  if (kids) {
    bocf_family_runs(kernel_id, kids, m, [&](int j0, int mr, int kid_) {
      launch_hypgrad(Xs + (long)j0 * strideXs, strideXs, N, Np, d, kid_, hyp + j0, alpha + (long)j0 * Np, Kinv + (long)j0 * strideK, strideK,
                     part + (long)j0 * hypgrad_num_blocks(Np) * (2 + d), out + (long)j0 * (2 + d), mr, s, nullptr, reduce);
    });
    return;
  }
  const int rows = hypgrad_rows(Np);
  dim3 grid((unsigned)((Np + 255) / 256), (unsigned)(Np / rows), (unsigned)m);
  const int kid = kernel_id <= 1 ? 0 : kernel_id;
#define LAUNCH(D, KID) BOCF_LAUNCH((hypgrad_kernel<D, KID>), grid, dim3(256), 0, s, Xs, strideXs, N, Np, hyp, alpha, Kinv, strideK, part, rows)
#define CASE(D)                       \
  case D:                             \
    if (kid == 0) LAUNCH(D, 0);       \
    else if (kid == 2) LAUNCH(D, 2);  \
    else LAUNCH(D, 3);                \
    break;
  switch (d) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
    CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29)
    CASE(30) CASE(31) CASE(32)
    default: break;
  }
#undef CASE
#undef LAUNCH
  if (reduce) BOCF_LAUNCH(hypgrad_reduce_kernel, dim3((unsigned)m), dim3(64), 0, s, part, hypgrad_num_blocks(Np), d, hyp, out);
}

// ---------------------------------------------------------------------------------------------
// Rank-1 append: the reference refits from scratch when cbo.py adds an observation
// (GP.set_XY, gp.py:191-227).  With U and R = U^-1 resident the bordered factor is
//   U' = [U u; 0 rho],  u = R^T k(X, x_new),  rho^2 = k(x,x) + noise + 1e-8 - ||u||^2
//   R' = [R -R u / rho; 0 1 / rho]
// i.e. two GEMVs and this O(N) write into row/column N of the identity padding.
__global__ __launch_bounds__(256) void append_write_kernel(double* __restrict__ S, double* __restrict__ R, double* __restrict__ RT,
                                                           long strideS, double* __restrict__ E, double* __restrict__ ET, long strideE,
                                                           int Np, int N, const double* __restrict__ u, const double* __restrict__ w,
                                                           const double* __restrict__ sumsq, long ldsumsq, const KernHyp* __restrict__ hyp,
                                                           int* __restrict__ fail) {
  const int j = blockIdx.y;
  const double diag = hyp[j].variance + hyp[j].noise + 1e-8;
  const double rho2 = diag - sumsq[(long)j * ldsumsq];
  // A new pivot inside the rounding noise of ||u||^2 (a near-duplicate observation with a tiny noise term: the difference
  // cancels to ~eps N diag) is not extended: the caller refits and the jitter ladder decides, as jitchol would from scratch.
  if (!(rho2 > 32.0 * 2.220446049250313e-16 * (double)(N + 1) * diag)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) fail[j] = 1;
    return;
  }
  const double rho = sqrt(rho2), irho = 1.0 / rho;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > N) return;
  double* __restrict__ Sj = S + (long)j * strideS;
  double* __restrict__ Rj = R + (long)j * strideS;
  double* __restrict__ RTj = RT + (long)j * strideS;
  const int p = N / NB, nl = N - p * NB;                 // diagonal block that holds row/column N
  double* __restrict__ Ej = E + (long)j * strideE + (long)p * NB * NB;
  double* __restrict__ ETj = ET + (long)j * strideE + (long)p * NB * NB;
  if (i == N) {
    Sj[(long)N * Np + N] = rho;
    Rj[(long)N * Np + N] = irho;
    RTj[(long)N * Np + N] = irho;
    Ej[nl * NB + nl] = irho;
    ETj[nl * NB + nl] = irho;
  } else {
    const double ui = u[(long)j * Np + i];
    const double ri = -w[(long)j * Np + i] * irho;
    Sj[(long)i * Np + N] = ui;          // U column N
    Sj[(long)N * Np + i] = ui;          // mirrored lower row (k-major view of U)
    Rj[(long)i * Np + N] = ri;          // R column N
    RTj[(long)N * Np + i] = ri;         // R^T row N
    if (i >= p * NB) {                  // the diagonal tile of R is E_p
      const int il = i - p * NB;
      Ej[il * NB + nl] = ri;
      ETj[nl * NB + il] = ri;
    }
  }
}

void launch_append_write(double* S, double* R, double* RT, long strideS, double* E, double* ET, long strideE, int Np, int N,
                         const double* u, const double* w, const double* sumsq, long ldsumsq, const KernHyp* hyp, int* fail, int m,
                         hipStream_t s) {
  BOCF_LAUNCH(append_write_kernel, dim3((unsigned)(N / 256 + 1), (unsigned)m), dim3(256), 0, s, S, R, RT, strideS, E, ET, strideE, Np, N,
                     u, w, sumsq, ldsumsq, hyp, fail);
}
