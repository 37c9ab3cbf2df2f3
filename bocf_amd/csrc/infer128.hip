// One whole hyper-parameter inference of a small model (N <= 128, d <= 16) in ONE workgroup per output, and the device-resident HMC chain
// built on it (exact_gaussian_inference.py:46-63, stationary.py:191-214; GPy/inference/mcmc/hmc.py:30-69).
#include "bocf_internal.h"
#include "fit_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

// ---------------------------------------------------------------------------------------------
// One whole hyper-parameter INFERENCE of a small model (N <= 128, d <= 16) in one workgroup per output: K(X,X) built
// straight into the register-resident 128x128 block, Cholesky, R = U^-1, alpha = R R^T yc, log-marginal, Ky^-1 = R R^T
// and the hyper-gradient reductions -- what bocf_fit + bocf_lml_gradients produce with ~15 launches, for the
// thousands of inferences of an HMC / optimiser update of the typical BO-sized model (gpmodel.py:115-118).
// Same arithmetic per element as the general path (build_train_kernel, chol128_regs / inv128_regs, hypgrad_kernel);
// only the order of the final sums differs.  out[j] = (d/dvariance, d/dnoise, d/dlengthscale_q ..., log-marginal, info):
// one small buffer, one device-to-host copy per inference.
#define INF_MAX_D 16
// Body of the fused inference as a device function (every thread of the 256-thread workgroup calls it; it ends with its results in
// out[0 .. 3 + d]: d/dvariance, d/dnoise, d/dlengthscale_q ..., log-marginal, info -- `out` may be global or LDS).  hj: this output's
// hyper-parameters (incl. the jitter of the current ladder rung); ycj: its centred targets (row of 128).  The stand-alone kernel calls
// it once; the resident HMC kernel (hmc128_kernel) once per leapfrog step.
// Phase stamps of ONE inference (probes build, env BOCF_DBG_ITS): s_memrealtime (100 MHz) of workgroup 0's first lane at the phase boundaries.
#ifdef BOCF_PROBES
__device__ unsigned long long g_its[16];
#define ITS(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_its[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ITS(k) do { } while (0)
#endif
template <int KID, int MFMA>
__device__ __forceinline__ void infer128_body(const double* __restrict__ X, int N, int d, const KernHyp* hj, const double* ycj, double* out) {
  __shared__ double rowbuf[2][NB];
  __shared__ double invd[NB];
  __shared__ double Ul[NB * 129];                        // U, later R = U^-1 (row stride 129)
  __shared__ double xs[NB * INF_MAX_D];                  // scaled inputs x_i / l, row stride d
  __shared__ double ycs[NB], tv[NB], al[NB];
  __shared__ double red[4][2 + INF_MAX_D];
  __shared__ int info_s;
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = tid & 15;
  const int nout = 2 + d + 2;
  ITS(0);
  if (tid == 0) info_s = 0;
  const double variance = hj->variance;
  const double dg = hj->noise + 1e-8 + hj->jitter;
  for (int idx = tid; idx < NB * d; idx += 256) {
    const int i = idx / d, q = idx - i * d;
    xs[idx] = i < N ? X[(long)i * d + q] / hj->ls[q] : 0.0;
  }
  if (tid < NB) ycs[tid] = tid < N ? ycj[tid] : 0.0;
  __syncthreads();
  ITS(1);
  const int nact = (N + 15) >> 4;                        // 16-row panels that hold real rows
  double sv = 0.0, sn = 0.0, sl[INF_MAX_D];              // hyper-gradient sums of this thread (variance, noise, lengthscales)
  if (MFMA) {
    // ---- K(X,X), Cholesky and R = U^-1 in ONE forward elimination of [Ky | I] on the matrix pipe: the body of potrf_diag_mfma_kernel
    // (16 x 16 tiles in the v_mfma_f64_16x16x4 accumulator layout, column block J = w + 4 jj per wave; per 16-row step: the owner wave
    // factors [D | I] inside one wave, the block row is multiplied by D^-T and published k-major, rank-16 update of the tiles below), with
    // K built straight into the tile registers (upper tiles only) and R = G^T written into the LDS image the rest of the inference
    // reads.  Only the ceil(N / 16) steps that hold real rows run: identity padding factors to itself.  (Round 3; the scalar
    // register-blocked phases below -- 95 of this kernel's 145 us at N = 128 -- are kept as MFMA = 0 for A/B.)
    const int lane_ = tid & 63, w_ = tid >> 6, c15 = lane_ & 15, q4 = lane_ >> 4;
    __shared__ double gdd[16][DD_LD];                    // G_dd of the current step: the A operand of the block-row product
    double (*pan)[16][PAN_LD] = reinterpret_cast<double (*)[16][PAN_LD]>(Ul);   // 69,632 of Ul's 132,096 bytes; R goes there afterwards
    static_assert(2 * 16 * PAN_LD <= NB * 129, "panel images alias the R image");
    double t[2][8][4], gd[2][4];
    double ex[2][8][4];                                  // the exponential factor of every kernel value (upper tiles): the hyper-gradient sums reuse it
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int J = w_ + 4 * jj;
#pragma unroll
      for (int I = 0; I < 8; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + 4 * r + q4, col = 16 * J + c15;
          double v = 0.0, e = 0.0;
          if (I <= J) {
            if (row < N && col < N) {
              double r2 = 0.0;
              for (int q = 0; q < d; ++q) {
                const double df = xs[row * d + q] - xs[col * d + q];
                r2 += df * df;
              }
              // (kern_of_r2 with the exponential kept)
              if (KID <= 1) {
                e = bocf_exp_nonpos(-0.5 * r2);
                v = variance * e;
              } else {
                const double rr = sqrt(r2);
                if (KID == 2) {
                  const double s5r = 2.23606797749978969641 * rr;
                  e = bocf_exp_nonpos(-s5r);
                  v = variance * (1.0 + s5r + (5.0 / 3.0) * r2) * e;
                } else {
                  const double s3r = 1.73205080756887729353 * rr;
                  e = bocf_exp_nonpos(-s3r);
                  v = variance * (1.0 + s3r) * e;
                }
              }
              if (row == col) v = variance + dg;
            } else {
              v = (row == col) ? 1.0 : 0.0;
            }
          }
          t[jj][I][r] = v;
          ex[jj][I][r] = e;
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) gd[jj][r] = (4 * r + q4 == c15) ? 1.0 : 0.0;
    }
    ITS(2);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      if (kb >= nact) break;                               // (uniform) identity padding factors to itself
      const int ow = kb & 3, oj = kb >> 2;
      double (*pn)[PAN_LD] = pan[kb & 1];
      if (kb == 0) ITS(9);
      if (w_ == ow) {                                      // (a) the owner wave factors [D | I] in its registers
        v4d_t dd = (v4d_t){t[oj][kb][0], t[oj][kb][1], t[oj][kb][2], t[oj][kb][3]};
        v4d_t gg = (v4d_t){gd[oj][0], gd[oj][1], gd[oj][2], gd[oj][3]};     // (the identity)
        chol16_regs(dd, gg, lane_, &info_s, 16 * kb);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t[oj][kb][r] = dd[r];
          gd[oj][r] = gg[r];
          gdd[4 * r + q4][c15] = gg[r];
          pn[4 * r + q4][128 + 16 * kb + c15] = gg[r];
        }
      }
      if (kb == 0) ITS(10);
      __syncthreads();
      if (kb == 0) ITS(11);
      double ga[4];                                        // (b) block row kb: X <- G_dd X, published k-major
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) ga[s4] = gdd[c15][4 * s4 + q4];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int J = w_ + 4 * jj;
        if (J != kb) {
          double x[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int I = 0; I < 8; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = I == kb ? t[jj][I][r] : x[r];
          v4d_t y = (v4d_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) y = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[s4], x[s4], y, 0, 0, 0);
          const int colbase = (J > kb ? 0 : 128) + 16 * J;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pn[4 * r + q4][colbase + c15] = y[r];
#pragma unroll
            for (int I = 0; I < 8; ++I) t[jj][I][r] = I == kb ? y[r] : t[jj][I][r];
          }
        }
      }
      if (kb == 0) ITS(12);
      __syncthreads();
      if (kb == 0) ITS(13);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {                     // (c) rank-16 update of the tiles below
        const int J = w_ + 4 * jj;
        double fb[4];
        const int bcol = (J > kb ? 0 : 128) + 16 * J;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) fb[s4] = pn[4 * s4 + q4][bcol + c15];
#pragma unroll
        for (int I = 1; I < 8; ++I) {
          if (I > kb && (I <= J || J <= kb)) {
            v4d_t acc = (v4d_t){t[jj][I][0], t[jj][I][1], t[jj][I][2], t[jj][I][3]};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              const double fa = -pn[4 * s4 + q4][16 * I + c15];
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb[s4], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) t[jj][I][r] = acc[r];
          }
        }
      }
      if (kb == 0) ITS(14);
    }
    __syncthreads();                                       // every read of the panel images is done: the G image takes their place
    ITS(3);
    // ---- G = U^-T (lower) as an LDS image, row stride 129 (every wave writes its column blocks); 1 / U_ii for the log-determinant
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int J = w_ + 4 * jj;
#pragma unroll
      for (int I = 0; I < 8; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = 4 * r + q4;
          Ul[(16 * I + rl) * 129 + 16 * J + c15] = I > J ? t[jj][I][r] : (I == J ? gd[jj][r] : 0.0);
          if (I == J && rl == c15) invd[16 * J + c15] = 1.0 / t[jj][I][r];
        }
    }
    __syncthreads();
    ITS(4);
    // ---- alpha = G^T (G yc) (exact_gaussian_inference.py:51), both products on the matrix pipe with the vector as column 0 of the B operand:
    // tv(I) = sum_{A <= I} G(I, A) yc(A)  (A operand G(I, A)[m][k] from the image),  alpha(A) = sum_{I >= A} G(I, A)^T tv(I)  (A operand = this
    // wave's own tile registers: the accumulator layout of a tile IS the A-operand layout of its transpose)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int I = w_ + 4 * jj;
      if (I < nact) {
        v4d_t acc = (v4d_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int A = 0; A <= I; ++A)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const double av = Ul[(16 * I + c15) * 129 + 16 * A + 4 * s4 + q4];
            const double bv = c15 == 0 ? ycs[16 * A + 4 * s4 + q4] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
          }
        if (c15 == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) tv[16 * I + 4 * r + q4] = acc[r];
        }
      } else if (c15 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tv[16 * I + 4 * r + q4] = 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int J = w_ + 4 * jj;
      v4d_t acc = (v4d_t){0.0, 0.0, 0.0, 0.0};
      if (J < nact) {
#pragma unroll
        for (int I = 0; I < 8; ++I)
          if (I >= J && I < nact) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              const double av = I == J ? gd[jj][s4] : t[jj][I][s4];
              const double bv = c15 == 0 ? tv[16 * I + 4 * s4 + q4] : 0.0;
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
          }
      }
      if (c15 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) al[16 * J + 4 * r + q4] = acc[r];
      }
    }
    __syncthreads();
    ITS(5);
    // ---- Ky^-1 = G^T G, upper tiles (A <= B) of this wave's column blocks B, on the matrix pipe: Kinv(A, B) = sum_{I >= B} G(I, A)^T G(I, B), the
    // B operand from the tile registers, the A operand from the registers (A == B) or the image; then the hyper-gradient sums (hypgrad_kernel) on
    // the accumulator tiles -- same positions as the kernel values built above, whose exponentials are still in registers
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int J = w_ + 4 * jj;
      const int col = 16 * J + c15;
      const double acol = al[col];
#pragma unroll
      for (int A = 0; A < 8; ++A) {
        if (A > J || J >= nact) {                        // (uniform) not an upper tile of the real block: nothing to add
#pragma unroll
          for (int r = 0; r < 4; ++r) ex[jj][A][r] = 0.0;
          continue;
        }
        v4d_t acc = (v4d_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int I = 0; I < 8; ++I)
          if (I >= J && I < nact) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              const double bv = I == J ? gd[jj][s4] : t[jj][I][s4];
              const double av = A == J ? bv : Ul[(16 * I + 4 * s4 + q4) * 129 + 16 * A + c15];
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
          }
        // per element: g = dL/dK (exact_gaussian_inference.py:61), the variance / noise sums, and 2 g f -- the weight of (dx_q)^2 in the
        // lengthscale sums -- left in the register of the exponential it was made from
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * A + 4 * r + q4;
          double w2 = 0.0;
          if (row < N && col < N && col >= row) {          // upper triangle of the real block; (row, col) stands for (col, row) too
            const double g = 0.5 * (al[row] * acol - acc[r]);
            if (row == col) {
              sn += g;
              sv += g;                                     // K_ii / variance = 1
            } else {
              const double e = ex[jj][A][r];
              double kv, f;
              if (KID <= 1) {
                kv = variance * e;
                f = kv;
              } else {
                double r2 = 0.0;
                for (int q = 0; q < d; ++q) {
                  const double df = xs[row * d + q] - xs[col * d + q];
                  r2 += df * df;
                }
                const double rr = sqrt(r2);
                if (KID == 2) {
                  const double s5r = 2.23606797749978969641 * rr;
                  kv = variance * (1.0 + s5r + (5.0 / 3.0) * r2) * e;
                  f = (5.0 / 3.0) * variance * (1.0 + s5r) * e;
                } else {
                  const double s3r = 1.73205080756887729353 * rr;
                  kv = variance * (1.0 + s3r) * e;
                  f = 3.0 * variance * e;
                }
              }
              const double g2 = 2.0 * g;
              sv += g2 * kv / variance;
              w2 = g2 * f;
            }
          }
          ex[jj][A][r] = w2;
        }
      }
    }
    // lengthscale sums, one input dimension at a time: sum over this lane's elements of w2 (dx_q)^2, reduced over the wave into red[w][2 + q]
#pragma unroll 1
    for (int q = 0; q < d; ++q) {
      double sq = 0.0;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int J = w_ + 4 * jj;
        if (J >= nact) continue;
        const double xc = xs[(16 * J + c15) * d + q];
#pragma unroll
        for (int A = 0; A < 8; ++A) {
          if (A > J) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double df = xs[(16 * A + 4 * r + q4) * d + q] - xc;
            sq += ex[jj][A][r] * (df * df);
          }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
      if (lane_ == 0) red[w_][2 + q] = sq;
    }
    ITS(6);
  }
  ITS(7);
  // ---- log-marginal pieces ride along in the block reduction: slot 0 of an extra pass
  double ld = 0.0, dt = 0.0;
  if (tid < N) {
    ld = -log(invd[tid]);                                // log U_ii
    dt = al[tid] * ycs[tid];
  }
  const int lane = tid & 63, w = tid >> 6;
  constexpr int NRED = MFMA ? 2 : 2 + INF_MAX_D;         // (the MFMA form has reduced its lengthscale sums into red[][2 + q] already)
  double vals[2 + INF_MAX_D];
  vals[0] = sv;
  vals[1] = sn;
#pragma unroll
  for (int q = 0; q < INF_MAX_D; ++q) vals[2 + q] = MFMA ? 0.0 : sl[q];
#pragma unroll
  for (int t = 0; t < NRED; ++t) {
    double v = vals[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[w][t] = v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ld += __shfl_xor(ld, o, 64);
    dt += __shfl_xor(dt, o, 64);
  }
  __syncthreads();
  if (lane == 0) { rowbuf[0][w] = ld; rowbuf[1][w] = dt; }
  __syncthreads();
  if (tid < 2 + d) {
    double sum = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    if (tid >= 2) sum /= hj->ls[tid - 2];            // differences were in scaled coordinates: (dx/l)^2 / l = dx^2 / l^3
    out[tid] = sum;
  }
  if (tid == 0) {
    const double logdet_half = ((rowbuf[0][0] + rowbuf[0][1]) + rowbuf[0][2]) + rowbuf[0][3];
    const double ya = ((rowbuf[1][0] + rowbuf[1][1]) + rowbuf[1][2]) + rowbuf[1][3];
    out[2 + d] = 0.5 * (-(double)N * 1.8378770664093454836 - 2.0 * logdet_half - ya);
    out[3 + d] = (double)info_s;
  }
  ITS(8);
}

// The same body behind a call: the resident chain below keeps its own state live around thousands of inferences; inlined there the body's
// 64 + 64 tile registers no longer fit next to it (240 B of scratch per lane, +30 us per inference at N = 64).
template <int KID, int MFMA>
__device__ __attribute__((noinline)) void infer128_call(const double* X, int N, int d, const KernHyp* hj, const double* ycj, double* out) {
  infer128_body<KID, MFMA>(X, N, d, hj, ycj, out);
}

template <int KID, int MFMA>
__global__ __launch_bounds__(256, 1) void infer128_kernel(const double* __restrict__ X, int N, int d, const KernHyp* __restrict__ hyp,
                                                          const double* __restrict__ yc_all, double* __restrict__ out) {
  const int jo = blockIdx.x;
  infer128_body<KID, MFMA>(X, N, d, hyp + jo, yc_all + (long)jo * NB, out + (long)jo * (2 + d + 2));
}

// ---------------------------------------------------------------------------------------------
// Device-resident HMC over the hyper-parameters of a small model (N <= 128, d <= 16): GPy/inference/mcmc/hmc.py:30-69 with M = I as
// GPModel.updateModel runs it (gpmodel.py:117-118: 200 draws x 20 leapfrog steps = 4000 inferences per output and update).  One
// workgroup per output runs its WHOLE chain in one launch: every leapfrog step is one infer128_body (kernel matrix, Cholesky, inverse,
// alpha, log-marginal, hyper-gradients) followed by the O(P) scalar part on lane 0 -- the Logexp transform of paramz (restated in
// bocf_amd/hyper.py, whose arithmetic this follows operation for operation), the Gamma priors (priors.py:264-330), the
// momentum / position updates, the Hamiltonian and the Metropolis test.  Momenta and uniforms are drawn by the HOST in the reference's
// RNG order and handed in.  The outputs' chains are independent (hyper.py advances them in lockstep only to batch the inferences).
// jitchol's ladder (linalg.py:52-71) runs inside the step.  status[j]: 0 = chain complete; i + 1 = a factorization failed (or the
// parameters left the positive domain) inside draw i and on_failure is "raise" -- the host raises LinAlgError like hmc.py would.
#define HMC_MAXP (2 + INF_MAX_D)

// (probes build) time of workgroup 0 between the marks of one evaluation, summed over the chain: [0] entry -> HTS(0) = the host-side step and
// the domain check, [1] the inference, [2] priors / transforms on lane 0, and the evaluation count
#ifdef BOCF_PROBES
__device__ unsigned long long g_hts[8];
#define HTS(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
    g_hts[k] += now_ - hts_last; hts_last = now_; if (k == 2) g_hts[3] += 1; } } while (0)
#else
#define HTS(k) do { } while (0)
#endif
template <int KID, int MFMA>
__global__ __launch_bounds__(256, 1) void hmc128_kernel(HmcArgs a) {
#ifdef BOCF_PROBES
  unsigned long long hts_last = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { g_hts[0] = g_hts[1] = g_hts[2] = g_hts[3] = 0; }
#endif
  __shared__ KernHyp hs;
  __shared__ double res[HMC_MAXP + 2];
  __shared__ double th[HMC_MAXP], tg[HMC_MAXP], tg_old[HMC_MAXP], x[HMC_MAXP], x_old[HMC_MAXP], pm[HMC_MAXP];
  __shared__ double obj_s, obj_old, jit_s;
  __shared__ double lpt[HMC_MAXP], ljt[HMC_MAXP], tgt[HMC_MAXP];     // per-parameter terms of one evaluation (lane k of wave 0 computes parameter k)
  __shared__ int kfree[HMC_MAXP];                                    // k of the kf-th free parameter
  __shared__ int failed_s, ffail_s, dom_s, stop_s, acc_s, div_s;
  __shared__ long long ninf_s;
  const int jo = blockIdx.x, tid = threadIdx.x;
  const int P = a.P, d = a.d, nls = a.nls;
  const int* fx = a.fixed + (long)jo * P;
  int Pf = 0;
  for (int k = 0; k < P; ++k) Pf += fx[k] ? 0 : 1;
  const double* yc = a.yc + (long)jo * NB;
  if (tid == 0) {
    int kf = 0;
    for (int k = 0; k < P; ++k) {
      th[k] = a.theta[(long)jo * P + k];
      if (!fx[k]) kfree[kf++] = k;
    }
    acc_s = 0; div_s = 0; stop_s = 0; ninf_s = 0;
  }
  __syncthreads();
  // objective = -(log-marginal + log-prior) and its gradient w.r.t. the optimizer array at th (hyper.py _objective_terms); every thread calls
  auto evaluate = [&]() {
    if (tid == 0) {
      bool ok = true;
      for (int k = 0; k < P; ++k) ok = ok && isfinite(th[k]) && (k == P - 1 ? th[k] >= 0.0 : th[k] > 0.0);
      dom_s = ok ? 1 : 0;
      if (ok) {
        hs.variance = th[0];
        for (int q = 0; q < BOCF_MAX_D; ++q) hs.ls[q] = q < d ? th[1 + (nls == 1 ? 0 : q)] : 1.0;
        hs.noise = th[P - 1];
        hs.ymean = 0.0;
        jit_s = 0.0;
        hs.jitter = -a.diag_shift;
      }
      ninf_s++;
    }
    __syncthreads();
    HTS(0);
    if (dom_s) {
      for (int attempt = 0;; ++attempt) {
        infer128_call<KID, MFMA>(a.X, a.N, d, &hs, yc, res);
        __syncthreads();
        if (res[3 + d] == 0.0 || attempt >= a.max_tries) break;
        if (tid == 0) {                                                // jitchol's ladder (linalg.py:52-71)
          const double diag_mean = hs.variance + hs.noise + 1e-8 - a.diag_shift;
          jit_s = jit_s == 0.0 ? diag_mean * 1e-6 : jit_s * 10.0;
          hs.jitter = jit_s - a.diag_shift;
        }
        __syncthreads();
      }
    }
    HTS(1);
    // priors, Logexp Jacobian and the gradient transform: parameter k on lane k of wave 0 (a dozen transcendental calls side by side instead
    // of one after the other), then lane 0 adds the terms up in the order hyper.py does
    if (tid < 64) {
      const bool ff = !dom_s || res[3 + d] != 0.0;
      if (!ff && tid < P) {
#pragma clang fp contract(off)
        const int k = tid;
        const double am1 = a.prior_a - 1.0;
        const double thk = th[k];
        const int fixed_k = fx[k];
        lpt[k] = a.prior_const + am1 * log(thk) - a.prior_b * thk;
        ljt[k] = fixed_k ? 0.0 : (thk > 36.0 ? thk : log(expm1(thk))) - thk;
        double g;
        if (k == 0) g = res[0];
        else if (k == P - 1) g = res[1];
        else if (nls == d) g = res[2 + (k - 1)];
        else {
          g = 0.0;
          for (int q = 0; q < d; ++q) g += res[2 + q];
        }
        const double em = expm1(thk);
        const double pg = (am1 / thk - a.prior_b) + (fixed_k ? 0.0 : 1.0 / em);
        tgt[k] = -(g + pg) * (thk > 36.0 ? 1.0 : -expm1(-thk));
      }
      wave_lds_fence();
      if (tid == 0) {
#pragma clang fp contract(off)
        bool bad = ff;
        double obj = 0.0;
        if (!ff) {
          double lp = 0.0, lj = 0.0;
          for (int k = 0; k < P; ++k) lp += lpt[k];
          for (int k = 0; k < Pf; ++k) lj += ljt[kfree[k]];
          obj = -res[2 + d] - (lp + lj);
          for (int k = 0; k < Pf; ++k) {
            const double t = tgt[kfree[k]];
            tg[k] = t;
            bad = bad || !isfinite(t);
          }
          bad = bad || !isfinite(obj);
        }
        if (bad) {
          obj = INFINITY;
          for (int k = 0; k < Pf; ++k) tg[k] = 0.0;
        }
        obj_s = obj;
        failed_s = bad ? 1 : 0;
        ffail_s = ff ? 1 : 0;
      }
    }
    __syncthreads();
    HTS(2);
  };
  auto set_free_from_x = [&](const double* xv) {                       // o.optimizer_array = x: param_array[free] = Logexp.f(x)
    int kf = 0;
    for (int k = 0; k < P; ++k)
      if (!fx[k]) th[k] = hmc_logexp_f(xv[kf++]);
  };
  evaluate();
  // a start that does not factorize even with jitter (or lies outside the positive domain): hmc.py:44 lets jitchol's LinAlgError out of the
  // first _computeH() -- with raise_on_failure the chain stops before its first draw
  if (a.raise_on_failure && ffail_s) {
    if (tid == 0) {
      stop_s = 1;
      a.status[jo] = 1;
    }
    __syncthreads();
  }
  const double half_log_2pi = 0.91893853320467274178;
#pragma unroll 1
  for (int i = 0; i < a.ns && !stop_s; ++i) {
    double H_old = 0.0;
    if (tid == 0) {
#pragma clang fp contract(off)
      const double* mi = a.mom + ((long)jo * a.ns + i) * P;
      double pp = 0.0;
      for (int k = 0; k < Pf; ++k) {
        pm[k] = mi[k];
        pp += pm[k] * pm[k];
      }
      H_old = obj_s + Pf * half_log_2pi + pp / 2.0;
      int kf = 0;
      for (int k = 0; k < P; ++k)
        if (!fx[k]) {
          x_old[kf] = hmc_logexp_finv(th[k]);
          x[kf] = x_old[kf];
          a.chains[((long)jo * a.ns + i) * P + kf] = th[k];
          ++kf;
        }
      obj_old = obj_s;
      for (int k = 0; k < Pf; ++k) tg_old[k] = tg[k];
    }
    int diverged = 0;
#pragma unroll 1
    for (int it = 0; it < a.iters; ++it) {
      if (tid < Pf) {                                                  // (free parameter kf on lane kf: independent updates)
#pragma clang fp contract(off)
        const double h = -a.eps / 2.0;
        const int k = tid;
        pm[k] += h * tg[k];
        x[k] += a.eps * pm[k];
        th[kfree[k]] = hmc_logexp_f(x[k]);
      }
      __syncthreads();
      evaluate();
      if (a.raise_on_failure && ffail_s) {                             // (uniform: ffail_s was written before evaluate's last barrier)
        if (tid == 0) {
          stop_s = 1;
          a.status[jo] = i + 1;
          set_free_from_x(x_old);                                      // leave the model where the draw started
        }
        break;
      }
      diverged |= failed_s;
      if (tid == 0) {
#pragma clang fp contract(off)
        const double h = -a.eps / 2.0;
        for (int k = 0; k < Pf; ++k) pm[k] += h * tg[k];
      }
    }
    __syncthreads();
    if (stop_s) break;
    if (tid == 0) {
#pragma clang fp contract(off)
      double pp = 0.0;
      for (int k = 0; k < Pf; ++k) pp += pm[k] * pm[k];
      const double H_new = obj_s + Pf * half_log_2pi + pp / 2.0;
      const double kk = H_old > H_new ? 1.0 : exp(H_old - H_new);
      if (!diverged && isfinite(H_new) && a.uni[(long)jo * a.ns + i] < kk) {
        int kf = 0;
        for (int k = 0; k < P; ++k)
          if (!fx[k]) a.chains[((long)jo * a.ns + i) * P + kf++] = th[k];
        acc_s++;
      } else {
        div_s += diverged;
        set_free_from_x(x_old);
        obj_s = obj_old;
        for (int k = 0; k < Pf; ++k) tg[k] = tg_old[k];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int k = 0; k < P; ++k) a.theta[(long)jo * P + k] = th[k];
    a.accepted[jo] = acc_s;
    a.diverged[jo] = div_s;
    if (!stop_s) a.status[jo] = 0;
    a.n_infer[jo] = ninf_s;
  }
}

void launch_hmc128(const HmcArgs& a, int kernel_id, int m, hipStream_t s, const int* kids) {
  if (kids) {
    bocf_family_runs(kernel_id, kids, m, [&](int j0, int mr, int kid_) {
      HmcArgs r = a;                                       // the run's slice of every per-output array
      r.yc = a.yc + (long)j0 * NB;
      r.theta = a.theta + (long)j0 * a.P;
      r.fixed = a.fixed + (long)j0 * a.P;
      r.mom = a.mom + (long)j0 * a.ns * a.P;
      r.uni = a.uni + (long)j0 * a.ns;
      r.chains = a.chains + (long)j0 * a.ns * a.P;
      r.accepted = a.accepted + j0;
      r.diverged = a.diverged + j0;
      r.status = a.status + j0;
      r.n_infer = a.n_infer + j0;
      launch_hmc128(r, kid_, mr, s, nullptr);
    });
    return;
  }
  const int kid = kernel_id <= 1 ? 0 : kernel_id;
#define LAUNCH(KID) BOCF_LAUNCH((hmc128_kernel<KID, 1>), dim3((unsigned)m), dim3(256), 0, s, a)
  if (kid == 0) LAUNCH(0);
  else if (kid == 2) LAUNCH(2);
  else LAUNCH(3);
#undef LAUNCH
#ifdef BOCF_PROBES
  if (getenv("BOCF_DBG_ITS")) {
    unsigned long long h[8];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_hts), sizeof(h));
    const double n = h[3] ? (double)h[3] : 1.0;
    fprintf(stderr, "hmc128 N=%d: per evaluation (%llu): leapfrog step + domain check %.2f us | inference call %.2f us | priors, transforms %.2f us\n", a.N, h[3],
            h[0] * 0.01 / n, h[1] * 0.01 / n, h[2] * 0.01 / n);
  }
#endif
}

void launch_infer128(const double* X, int N, int d, int kernel_id, const KernHyp* hyp, const double* yc, double* out, int m, hipStream_t s,
                     const int* kids) {
  if (kids) {
    bocf_family_runs(kernel_id, kids, m, [&](int j0, int mr, int kid_) {
      launch_infer128(X, N, d, kid_, hyp + j0, yc + (long)j0 * NB, out + (long)j0 * (2 + d + 2), mr, s, nullptr);
    });
    return;
  }
  const int kid = kernel_id <= 1 ? 0 : kernel_id;
#define LAUNCH(KID) BOCF_LAUNCH((infer128_kernel<KID, 1>), dim3((unsigned)m), dim3(256), 0, s, X, N, d, hyp, yc, out)
  if (kid == 0) LAUNCH(0);
  else if (kid == 2) LAUNCH(2);
  else LAUNCH(3);
#undef LAUNCH
#ifdef BOCF_PROBES
  if (getenv("BOCF_DBG_ITS")) {
    static int shown = 0;
    if (shown++ % 1000 == 5) {
      unsigned long long h[16];
      (void)hipStreamSynchronize(s);
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_its), sizeof(h));
      static const char* nm[8] = {"stage x,y", "K build", "elimination", "R image", "alpha", "Kinv = R R^T", "gradient sums", "reductions"};
      fprintf(stderr, "infer128 N=%d d=%d:", N, d);
      for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.2f us |", nm[k], (double)(h[k + 1] - h[k]) * 0.01);
      fprintf(stderr, " total %.2f us", (double)(h[8] - h[0]) * 0.01);
      fprintf(stderr, " || step 0: factor %.2f, barrier %.2f, multiply %.2f, barrier %.2f, update %.2f us\n", (double)(h[10] - h[9]) * 0.01, (double)(h[11] - h[10]) * 0.01,
              (double)(h[12] - h[11]) * 0.01, (double)(h[13] - h[12]) * 0.01, (double)(h[14] - h[13]) * 0.01);
    }
  }
#endif
}

