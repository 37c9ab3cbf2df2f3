// Internal declarations shared by the HIP translation units of libbocf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BOCF_TILE 128          // panel width NB == GEMM tile edge; every matrix is padded to it
#define BOCF_MAX_D 32          // max input dimension
#define BOCF_MAX_M 16          // max model outputs (per hyper-sample)
#define BOCF_MAX_FITS 1024     // max independent factorizations in one fit (hyper-samples x outputs)
#define BOCF_MAX_L 32          // max utility-parameter support size on device

// Every kernel launch goes through BOCF_LAUNCH: the launch status (bad grid / block / shared-memory configuration) is read
// right behind the launch and the FIRST failure is remembered with the kernel's name; the C-ABI entry point that issued it
// reports it (bocf_last_error() names the kernel, not whichever call happened to synchronise next).
void bocf_note_launch(const char* kernel, hipError_t e);
#define BOCF_LAUNCH(kern, grid, block, shm, stream, ...)              \
  do {                                                                \
    hipLaunchKernelGGL(kern, grid, block, shm, stream, __VA_ARGS__);  \
    bocf_note_launch(#kern, hipGetLastError());                       \
  } while (0)

// s + c += a * b with the running sum carried as an unevaluated pair (s, c): TwoProd through fma, TwoSum (Knuth).  The posterior
// mean sum_k K(x*, X_k) alpha_k is a sum of N terms of size |alpha| ~ 1e6 (cond(Ky) ~ 4e9 at BASELINE configs[2]) that cancels to O(1):
// plain fp64 accumulation leaves 4e-8 absolute there, the pair 5e-9 (measured against oracle/truth_ld.c) -- below what the fp64
// rounding of K itself leaves.  (Contraction is switched off locally: fusing a * b into the following add would defeat TwoProd.)
__device__ __forceinline__ void dd_fma_acc(double& s, double& c, double a, double b) {
#pragma clang fp contract(off)
  const double p = a * b;
  const double e = __builtin_fma(a, b, -p);
  const double t = s + p;
  const double z = t - s;
  c += ((s - (t - z)) + (p - z)) + e;
  s = t;
}
__device__ __forceinline__ void dd_add_acc(double& s, double& c, double hi, double lo) {
#pragma clang fp contract(off)
  const double t = s + hi;
  const double z = t - s;
  c += ((s - (t - z)) + (hi - z)) + lo;
  s = t;
}

// exp(x) for x <= 0 down to the underflow threshold: the operation sequence of the device library's double-precision exp (argument reduction
// by rint(x log2 e) with a two-part ln 2, degree-11 polynomial, ldexp) WITHOUT its two range selects (x > 1024 -> inf, x < -1075 -> 0):
// every covariance function of the path takes exp of -r^2/2, -sqrt(5) r or -sqrt(3) r.  Same operations on the same operands: the same bits
// as exp() on 2^30 arguments across [-745.2, 0] (tools/hbm_kernel_probe.hip) and 0 below -1075; 5 of the ~56 vector instructions per element
// of the K builds.
__device__ __forceinline__ double bocf_exp_nonpos(double x) {
  x = __builtin_fmax(x, -1100.0);                        // (far-apart points with tiny lengthscales: below -1075 the library returns 0, and so does ldexp here)
  const double dn = __builtin_rint(x * 0x1.71547652b82fep+0);
  const double t = __builtin_fma(-dn, 0x1.abc9e3b39803fp-56, __builtin_fma(-dn, 0x1.62e42fefa39efp-1, x));
  double p = __builtin_fma(t, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
  p = __builtin_fma(t, p, 0x1.71dee623fde64p-19);
  p = __builtin_fma(t, p, 0x1.a01997c89e6b0p-16);
  p = __builtin_fma(t, p, 0x1.a01a014761f6ep-13);
  p = __builtin_fma(t, p, 0x1.6c16c1852b7b0p-10);
  p = __builtin_fma(t, p, 0x1.1111111122322p-7);
  p = __builtin_fma(t, p, 0x1.55555555502a1p-5);
  p = __builtin_fma(t, p, 0x1.5555555555511p-3);
  p = __builtin_fma(t, p, 0x1.000000000000bp-1);
  p = __builtin_fma(t, p, 1.0);
  p = __builtin_fma(t, p, 1.0);
  return __builtin_ldexp(p, (int)dn);
}

// Outputs may use different kernel FAMILIES (the reference's multi_outputGP takes a kernel list, multi_outputGP.py:44-47).  The kernels
// that evaluate a covariance function are specialised per family at compile time, so a launcher that is given `kids` (host array of m
// kernel ids, or nullptr = every output uses `kernel_id`) issues one launch per RUN of equal ids, with its pointers advanced to the
// run's first output: f(j0, m_run, kernel_id_of_the_run).
template <typename F>
static inline void bocf_family_runs(int kernel_id, const int* kids, int m, F f) {
  if (!kids) {
    f(0, m, kernel_id);
    return;
  }
  for (int j0 = 0; j0 < m;) {
    int j1 = j0 + 1;
    while (j1 < m && kids[j1] == kids[j0]) ++j1;
    f(j0, j1 - j0, kids[j0]);
    j0 = j1;
  }
}

// ---------------------------------------------------------------------------------------
// f64 MFMA GEMM (gemm_f64.hip):  C[r][c] = beta*Cin[r][c] + alpha * sum_kk A[kk][r] * B[kk][c]
// A and B are stored k-major (the contraction index is the slow one), C is row-major.
// Per-tile contraction length: kend = min(K, kb + krt*rt + kct*ct)  (rt/ct = tile row/col).
// ---------------------------------------------------------------------------------------
struct GemmArgs {
  const double* A;  long lda;  long strideA;   // batch stride (per blockIdx.z)
  const double* B;  long ldb;  long strideB;
  const double* Cin; double* Cout; long ldc; long strideC;
  int M, Ncols, K;             // multiples of 128 / 128 / 16
  int kb, krt, kct;            // per-tile contraction length rule
  int kbeg_rt, kbeg_ct;        // contraction starts at kbeg_rt * rt + kbeg_ct * ct (lower-triangular operands)
  int batch1;                  // blockIdx.z = z2 * batch1 + z1; second-level strides below (0 = unused)
  long strideA2, strideB2, strideC2;
  int upper_only;              // skip tiles with ct < rt (symmetric rank-k update)
  int rt_desc;                 // schedule heavy (large rt) tiles first
  int ct_desc;                 // idem for contraction lengths that grow with the column tile (kct > 0)
  int swizzle;                 // 0 plain order; 2 row-tile-major across the batch (heaviest rows first); 258 the 256-row variance kernel; probes build: 1, 100 + RT, 256, 257
  int batch;                   // set by the launcher
  int prefetch1;               // A/B switch: 1 = one-tile-deep staging in the sumsq variant
  int stagger;                 // probes build, two-buffer 256-row kernel: waves 4..7 store their operand share before the MFMAs
  int no_x3;                   // 1: never route a triangular store product to the three-buffer kernel (A/B, tests)
  int vprobe;                  // timing-only: the 256-row variance kernel also executes the VALU work of a fused K* tile build
  double alpha, beta;
  double* sumsq;               // epilogue 1: partial column sums of squares [batch][rt][Ncols]
  long strideSumsq;            // batch stride of sumsq
};
// epilogue 0: store C;  epilogue 1: write sumsq partials only (C is never stored)
void launch_gemm_f64(const GemmArgs& g, int batch, int epilogue, hipStream_t s);

// fp32 variance contraction (gemm_f32.hip): sum-of-squares epilogue only
struct GemmArgs32 {
  const float* A; long lda; long strideA;
  const float* B; long ldb; long strideB;
  int M, Ncols, K;
  double* sumsq; long strideSumsq;
  int tile128;                 // 1: always the 128-row kernel (option "swizzle" = 0: A/B, tests)
};
void launch_gemm_f32_sumsq(const GemmArgs32& g, int batch, hipStream_t s);
void launch_f64_to_f32(const double* src, float* dst, long n, hipStream_t s);

// ---------------------------------------------------------------------------------------
// fit kernels (fit.hip)
// ---------------------------------------------------------------------------------------
struct KernHyp {               // per-output hyper-parameters, device-resident array of these
  double variance;
  double noise;
  double ymean;
  double ls[BOCF_MAX_D];       // lengthscale_q (inputs are DIVIDED by it, as the reference does)
  double jitter;               // diagonal jitter of the current ladder attempt (read by the fused inference kernel only)
};

void launch_scale_inputs(const double* X, int n, int d, const KernHyp* hyp, int m, double* Xs, long strideXs, hipStream_t s);
// K(X,X) -> S (Np x Np per output, upper tiles), diag += noise + 1e-8 + jitter[j]; padding = identity
void launch_build_train_kernel(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp,
                               const double* jitter, int add_diag, double* S, long strideS, int m, hipStream_t s, const int* kids = nullptr);
// factor the p-th 128x128 diagonal block in place (upper, A = U^T U), write E = U^-1 and E^T
// one 128 x 128 tile per output, K = 128: C = beta C + alpha A^T B (A, B k-major), sixteen 32 x 32 pieces, a wave each
// (ntiles tiles side by side: B and C advance by 128 columns per tile, A is shared)
void launch_tile128(const double* A, long lda, long strideA, const double* B, long ldb, long strideB, double* C, long ldc, long strideC,
                    double alpha, double beta, int m, hipStream_t s, int ntiles = 1, int K = 128, int* done = nullptr);
#ifdef BOCF_PROBES
void dbg_tl_start();                     // debug timeline of the factorization's chain kernels (env BOCF_DBG_TL, probes build)
void dbg_tl_dump(const char* path);
#endif
void set_potrf_scalar(int on);   // 1: scalar diagonal-block kernel instead of the MFMA form (process-wide A/B switch)
// done (optional): device counter the kernel's workgroups add 1 to when their output is released (dependencies across streams)
void launch_potrf_diag(double* S, long strideS, int N, int Np, int p, double* E, double* ET, long strideE, int* info, int m, hipStream_t s,
                       int* done = nullptr);
// single-wave launch that polls up to two device counters until they reach n0 / n1 (bounded; *err = 1 on a timeout)
void launch_gate(const int* f0, int n0, const int* f1, int n1, int* err, hipStream_t s);
void launch_signal(int* f, int add, hipStream_t s);
// One-launch factorization + inverse of models with few panels (chol_team.hip): every output has a team of T resident workgroups
struct TeamArgs {
  double* S; double* RT; double* R; long strideS;   // upper factor (in: Ky, out: U), R^T (lower) and R (upper), Np x Np per output
  double* E; double* ET; long strideE;              // inverted diagonal blocks
  int N, Np, nb;
  int* info;                                        // per output: LAPACK-style info of the diagonal blocks
  int* F; int fstride;                              // per-output counters (chol_team_flag_words(nb) ints each), zeroed by the caller
  int* err;                                         // first wait that ran out of polls (0 = none)
  int T;                                            // workgroups per team (>= 2)
  int p0, p1;                                       // panels [p0, p1) of the factorization (the whole of it: 0, nb; do_inverse needs that)
  int do_inverse;
  double* KI; int do_kinv;                          // with do_inverse: also Ky^-1 = R R^T (upper tiles, Np x Np per output, stride strideS)
  int crit_load;                                    // the workgroups of the critical units carry nothing else while the others get by with <= this many units each
  int stream;                                       // a workgroup of the team streams the critical tiles underneath the diagonal blocks (team_crit_stream)
  unsigned long long* tl;                           // probes build: per-workgroup task timeline (nullptr = off)
};
#define TEAM_MAX_NB 32
int chol_team_flag_words(int nb);
void launch_chol_team(const TeamArgs& a, int m, hipStream_t s);
// whole inference (log-marginal + hyper-gradients) of a model with N <= 128, d <= 16 in one launch; yc has row stride 128
#define BOCF_INFER_MAX_D 16
// out: m rows of (2 + d gradients, log-marginal, info)
void launch_infer128(const double* X, int N, int d, int kernel_id, const KernHyp* hyp, const double* yc, double* out, int m, hipStream_t s,
                     const int* kids = nullptr);
// device-resident HMC chain of a small model (fit.hip hmc128_kernel): one workgroup per output runs the whole chain
struct HmcArgs {
  const double* X; int N, d; const double* yc;      // yc: (m, 128) centred targets
  double* theta; const int* fixed; int P, nls;      // theta (m, P) in/out: [variance, lengthscale (nls = 1 or d), noise]
  double prior_a, prior_b, prior_const;             // Gamma(a, b): lnpdf = const + (a - 1) log x - b x
  const double* mom; const double* uni;             // (m, ns, P) free entries packed in front; (m, ns)
  int ns, iters; double eps; int max_tries, raise_on_failure; double diag_shift;
  double* chains; int* accepted; int* diverged; int* status; long long* n_infer;
};
void launch_hmc128(const HmcArgs& a, int kernel_id, int m, hipStream_t s, const int* kids = nullptr);
// stream-resident HMC chain of larger models (hmc_stream.hip): the O(P) arithmetic between the inferences of the leapfrog steps
enum { HS_INIT = 0, HS_EVAL0 = 1, HS_PRE = 2, HS_POST = 3, HS_FINISH = 4 };
struct HmcStreamArgs {
  int m, P, nls, d, ns, iters;
  double eps, prior_a, prior_b, prior_const, diag_shift;
  const int* fixed;                                 // (m, P)
  const double* mom; const double* uni;             // (m, ns, P) free entries packed in front; (m, ns)
  double* state;                                    // hmc_stream_state_doubles(m)
  double* chains; int* accepted; int* diverged; long long* n_eval;
  int* abort_draw;                                  // -1, or the first draw the host has to run (jitter ladder / domain exit / schedule time-out)
  KernHyp* hyp;                                     // the hyper-parameters the inference's kernels read
  // the inference's own small launches folded into PRE / POST: inputs scaled by the lengthscales, zeroed schedule counters (PRE);
  // log-marginal from diag(U), alpha, yc and the reduction of the hyper-gradient partials (POST)
  const double* X; double* Xs; long strideXs; int N, Np;
  int* flags; int flag_words;                       // per-output counters of the team schedule (+ its time-out word behind them), or nullptr
  const double* S; long strideS; const double* alpha; const double* yc;
  const double* part; int nblk;                     // (m, nblk, 2 + d) partials of hypgrad_kernel: [dvariance, dnoise, dlengthscale ...]
  int* info; const int* sched_err;                  // per-output pivot status of the factorization; time-out word of its schedule (or nullptr)
  double* theta;                                    // (m, P) in / out
};
int hmc_stream_state_doubles(int m);
void launch_hmc_stream(const HmcStreamArgs& a, int mode, int i, int it, hipStream_t s);
// copy the diagonal 128x128 blocks [blk_lo, blk_hi) of E into the diagonal tiles of R
void launch_copy_diag_blocks(const double* E, long strideE, double* R, long strideR, int Np, int blk_lo, int blk_hi, int m, hipStream_t s);
// dst[(c0+c)][(r0+r)] = src[(r0+r)][(c0+c)] for a rows x cols block, `count` blocks spaced `step` along the diagonal
void launch_transpose_block(const double* src, double* dst, long stride, int Np, int r0, int c0, int rows, int cols, int count, int step,
                            int m, hipStream_t s);
// tiles on / above the diagonal of one Np x Np matrix <-> a contiguous buffer of nb (nb + 1) / 2 tiles (the exchange format of the sharded fit)
void launch_pack_upper_tiles(const double* R, int Np, double* packed, hipStream_t s);
void launch_unpack_upper_tiles(const double* packed, int Np, double* R, hipStream_t s);
// alpha = R t (t = R^T y comes from launch_gemv_small_t)
void launch_gemv_upper_n(const double* R, long strideR, int Np, const double* t, double* alpha, int m, hipStream_t s);
// Rank-1 append of one observation (row/column N of the padded factor; requires N < Np).  u = R^T k_new (Np),
// w = R u (Np), sumsq = ||u||^2 per output.  On a non-positive pivot fail[j] = 1 and nothing is written.
void launch_append_write(double* S, double* R, double* RT, long strideS, double* E, double* ET, long strideE, int Np, int N,
                         const double* u, const double* w, const double* sumsq, long ldsumsq, const KernHyp* hyp, int* fail, int m,
                         hipStream_t s);

// d lml / d (variance, lengthscale_q, noise) per output from alpha and Kinv (upper tiles valid):
// part: (m, nblocks, 2 + d) scratch; out: (m, 2 + d) = [dvariance, dnoise, dls_0 ... dls_{d-1}]
int hypgrad_num_blocks(int Np);
void launch_hypgrad(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* alpha,
                    const double* Kinv, long strideK, double* part, double* out, int m, hipStream_t s, const int* kids = nullptr, bool reduce = true);
void launch_lml(const double* S, long strideS, int N, int Np, const double* alpha, const double* yc, double* lml, int m, hipStream_t s);
// One step of iterative refinement of alpha = Ky^-1 yc with the residual in double-double (exact_gaussian_inference.py:51 solves once
// with dpotrs; at cond(Ky) ~ 4e9 that -- like R (R^T yc) here -- leaves ~4e-8 relative in alpha, the refined alpha 2e-9):
//   launch_kalpha_dd:   part[j][blk][0/1][i] = the pair (hi, lo) of sum_{k in 128-block blk} Ky[i][k] alpha[k], Ky rebuilt on the fly
//                       exactly as build_train_kernel stores it (diagonal = variance + (noise + 1e-8 + jitter))
//   launch_refine_rhs:  r[j][i] = yc[i] - sum_blk pairs      (rows >= N: 0)
//   (the caller solves delta = R (R^T r) with the GEMV kernels)
//   launch_refine_apply: alpha += delta;  mu_train[j][i] = yc[i] + ymean - dg alpha[i]   (K alpha = yc - dg alpha for the solution of
//                       (K + dg I) alpha = yc: the posterior mean at the training inputs, multi_outputGP.py:176-180, without another
//                       pass over K and without the cancellation of the direct sum)
void launch_kalpha_dd(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* jitter,
                      const double* alpha, double* part, int m, hipStream_t s, const int* kids = nullptr);
// int8 (Ozaki) form of the variance contraction (gemm_i8.hip, option "predict_i8"): digit fragments of R (per fit) and of K* (per chunk),
// exact int8 products, fp64 recombination, the per-128-row sums of squares of the fp64 kernels' layout
#define BOCF_I8_SLICES 6
size_t i8_operand_bytes(int Np, int ncols, int m);        // digit fragments of an Np x ncols operand (ncols a multiple of 16), m matrices
void launch_col_exponents(const double* R, long strideR, int Np, int* expo, int m, hipStream_t s);
// X[k][col] (k-major, leading dimension ld) -> fragments F; expo: per column (expo_stride = columns per matrix) or one per matrix (0)
void launch_slice_operand(const double* X, long ld, long strideX, int krows, int Np, int ncols, const int* expo, int expo_stride, void* F, int m,
                          hipStream_t s);
void launch_var_i8(const void* Af, const void* Bf, int Np, int ncols, const int* eA, const int* eB, double* sumsq, long strideSumsq, int m,
                   hipStream_t s, int group = 0);
int kalpha_block(int Np);                              // columns per partial sum of launch_kalpha_dd (part holds Np / that many pairs per row)
void launch_refine_rhs(const double* part, int N, int Np, const double* yc, double* r, int m, hipStream_t s);
void launch_refine_apply(const double* delta, int N, int Np, const KernHyp* hyp, const double* jitter, const double* yc, double* alpha,
                         double* mu_train, long ldmu, int m, hipStream_t s);

// ---------------------------------------------------------------------------------------
// predict kernels (predict.hip)
// ---------------------------------------------------------------------------------------
// cross kernel K*[j][kk][c] (Np x ldk per output; rows >= N zero) + partial means
void launch_cross_kernel(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp,
                         const double* Xc, int c0, int Cn, int Cpad, const double* alpha, double* Kstar, long ldk, long strideK,
                         double* meanpart, double* meanlo, int nsplit, int m, int store_k, hipStream_t s, const int* kids = nullptr);
// (meanpart / meanlo: the partial means per 128-row block as unevaluated pairs hi + lo, same layout each)
// store_k: 0 = mean only, 1 = K* as fp64, 2 = K* as fp32 (Kstar then points to float storage; ldk/strideK in elements)
void launch_finalize_mean(const double* meanpart, const double* meanlo, int nsplit, int Cpad, const KernHyp* hyp, double* mean, long ldmean, int c0,
                          int Cn, int m, hipStream_t s);
// (meanpart != nullptr: the same launch finishes the means too -- mean and var share ldvar)
void launch_finalize_var(const double* sumsq, int nrt, int Cpad, const KernHyp* hyp, int flags, double* var, long ldvar, int c0, int Cn, int m, hipStream_t s,
                         const double* meanpart = nullptr, const double* meanlo = nullptr, int nsplit = 0, double* mean = nullptr);
// column 0 of the predictive covariance from the mean-shaped pass t (see cov_column_kernel)
void launch_cov_column(const double* Xc, int C, int d, int kernel_id, const KernHyp* hyp, const double* t, long ldt, int flags, double* cov,
                       long ldcov, int m, hipStream_t s, const int* kids = nullptr);

// Low-latency path for n <= BOCF_SMALL_N candidates (single-point L-BFGS calls): GEMV-shaped, R read once.
#define BOCF_SMALL_N 16
// V[j][r][c] = sum_{kk<=r} R[kk][r] K*[kk][c]   (V: Np x nc per output, nc in {1,2,4,8,16}; K*: ldk columns)
void launch_gemv_small_t(const double* R, long strideR, int Np, const double* Kstar, long ldk, long strideK, double* V, int nc, int m, hipStream_t s);
// sumsq[j][c] = sum_r V[r][c]^2   (written as a single row-tile partial: nrt = 1, row length ldo)
void launch_sumsq_small(const double* V, int Np, double* sumsq, long ldo, int nc, int m, hipStream_t s);
// W[j][r][c] = sum_{kk>=r} R[r][kk] V[kk][c]   (W: Np x 16 per output)
void launch_gemv_small_n(const double* R, long strideR, int Np, const double* V, double* W, int nc, int m, hipStream_t s);
// the same three steps on the matrix pipe (predict.hip): K* + mean partials of <= 16 candidates; V = R^T K* with the column sums of squares per
// 16-row tile (sumsq[j][Np / 16][ldss]); W = R V from R^T
void launch_cross_small(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* Xc, int c0, int Cn, int nc,
                        const double* alpha, double* Kstar, long ldk, long strideK, double* meanpart, double* meanlo, int Cpad, int m, hipStream_t s,
                        const int* kids = nullptr);
void launch_gemv_small_t_mfma(const double* R, long strideR, int Np, const double* Kstar, long ldk, long strideK, double* V, double* sumsq, long ldss,
                              int nc, int m, hipStream_t s);
void launch_gemv_small_n_mfma(const double* RT, long strideR, int Np, const double* V, double* W, int nc, int m, hipStream_t s);

// d mean / dx and d var / dx of every candidate: (m, ldg, d) each.  W = Ky^-1 K(X, X*) (Np x ldw per output).
void launch_grad_kernel(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* Xc, int c0,
                        int Cn, const double* alpha, const double* W, long ldw, long strideW, double* dmean, double* dvar, long ldg,
                        int m, hipStream_t s, const int* kids = nullptr);

// ---------------------------------------------------------------------------------------
// acquisition + selection kernels (acq.hip)
// ---------------------------------------------------------------------------------------
struct AcqArgs {
  const double* mean; const double* var; long ld;   // (m, ld), first C columns valid
  int m, C, L;
  int kind;                // BOCF_ACQ_*
  int util_kind;           // BOCF_UTIL_*
  int theta_dim;
  const double* theta;     // device (L, theta_dim)
  const double* prob;      // device (L) weights (already 1/L when the reference takes a plain mean)
  const double* best;      // device (L)
  const double* util_params; int n_util_params;
  const double* Wt; int S; // device (m, S) transposed normals
  double* acq;             // device (C)
  const double* dmean; const double* dvar; long ldg; int d;   // gradient variants: (m, ldg, d)
  double* dacq;            // device (C, d)
  int accumulate;          // 0: acq/dacq are written; 1: added to (hyper-sample h > 0 of the h-loop, maEI.py:85-97)
  double scale;            // 1 / H
};
void launch_best_so_far(const double* mu_train, int N, int m, int linear, int util_kind, const double* theta, int theta_dim, int L,
                        const double* util_params, double* best, hipStream_t s);
void launch_acq_linear(const AcqArgs& a, hipStream_t s);
void launch_acq_mc(const AcqArgs& a, hipStream_t s);
void launch_acq_linear_grad(const AcqArgs& a, hipStream_t s);
void launch_acq_mc_grad(const AcqArgs& a, hipStream_t s);
// two-stage top-k (value desc, index asc); out: idx (k) int64, val (k)
void launch_topk(const double* acq, int C, int k, long long* blk_idx, double* blk_val, long long* out_idx, double* out_val, hipStream_t s);
int topk_num_blocks(int C);
// multi-GPU selection: pack the k local winners (+ lo) into a 2 * world * k buffer of doubles; merge the gathered buffer
void launch_pack_topk(const long long* idx, const double* val, int k, long long lo, int world, int rank, double* pack, hipStream_t s);
void launch_merge_packed(const double* pack, int k, int world, long long* gidx, double* gval, long long* out_idx, double* out_val, hipStream_t s);
