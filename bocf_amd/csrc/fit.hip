// Fit-side kernels: K(X,X) build, diagonal-block Cholesky + inverse, single-tile products, transposes, GEMVs, log-marginal.
// Storage convention: every N x N matrix is padded to Np (multiple of 128) and held ROW-MAJOR in
// its UPPER form: Ky = U^T U with U upper triangular (U = L^T of the reference's lower factor,
// GPy/util/linalg.py:52-55), R = U^-1 upper.  Padding rows/cols carry the identity.
#include "bocf_internal.h"
#include "fit_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

static thread_local int g_potrf_scalar = 0;     // (set by the launching thread right before its launches: contexts on different threads do not race) option "potrf_scalar": 0 = MFMA form with a factor wave; 1 = scalar register-blocked kernel; 2 = round-2a MFMA form (A/B, tests)

// ---------------------------------------------------------------------------------------------

// (producer half of the device-side dependencies described at gate_kernel below)
__device__ __forceinline__ void dep_signal(int* done) {
  if (!done) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Debug timeline (env BOCF_DBG_TL=<file>, read by the factorization schedules in capi.hip): the latency-chain kernels of this file stamp
// s_memrealtime (100 MHz) at entry and exit of every workgroup into a device buffer -- what a multi-stream schedule does in a PLAIN run
// (rocprofv3's kernel trace changes the timing of exactly these schedules).
__device__ unsigned long long* g_dbg_tl = nullptr;       // [0] = record count, then records of 4 words: id, block, t0, t1
#define DBG_TL_MAX 400000
__device__ __forceinline__ unsigned long long tl_begin() { return g_dbg_tl ? (unsigned long long)__builtin_amdgcn_s_memrealtime() : 0ull; }
__device__ __forceinline__ void tl_end(int id, unsigned long long t0) {
  if (g_dbg_tl && threadIdx.x == 0) {
    const unsigned long long t1 = (unsigned long long)__builtin_amdgcn_s_memrealtime();
    const unsigned long long idx = atomicAdd(&g_dbg_tl[0], 1ull);
    if (idx < DBG_TL_MAX) {
      unsigned long long* r = g_dbg_tl + 1 + 4 * idx;
      r[0] = (unsigned long long)id;
      r[1] = (unsigned long long)blockIdx.x | ((unsigned long long)blockIdx.y << 32);
      r[2] = t0;
      r[3] = t1;
    }
  }
}
static unsigned long long* g_dbg_tl_host = nullptr;
void dbg_tl_start() {
  if (!g_dbg_tl_host) {
    if (hipMalloc(&g_dbg_tl_host, sizeof(unsigned long long) * (1 + 4 * (size_t)DBG_TL_MAX)) != hipSuccess) return;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_tl), &g_dbg_tl_host, sizeof(g_dbg_tl_host));
  }
  (void)hipMemset(g_dbg_tl_host, 0, sizeof(unsigned long long));
}
void dbg_tl_dump(const char* path) {
  if (!g_dbg_tl_host) return;
  (void)hipDeviceSynchronize();
  unsigned long long n = 0;
  (void)hipMemcpy(&n, g_dbg_tl_host, sizeof(n), hipMemcpyDeviceToHost);
  if (n > DBG_TL_MAX) n = DBG_TL_MAX;
  std::vector<unsigned long long> rec(4 * n);
  if (n) (void)hipMemcpy(rec.data(), g_dbg_tl_host + 1, sizeof(unsigned long long) * 4 * n, hipMemcpyDeviceToHost);
  FILE* f = fopen(path, "w");
  if (!f) return;
  for (unsigned long long i = 0; i < n; ++i)
    fprintf(f, "%llu %llu %llu %llu %llu\n", rec[4 * i], rec[4 * i + 1] & 0xffffffffull, rec[4 * i + 1] >> 32, rec[4 * i + 2], rec[4 * i + 3]);
  fclose(f);
}


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
__global__ __launch_bounds__(256) void build_train_kernel(const double* __restrict__ Xs, long strideXs, int N, int Np,
                                                          const KernHyp* __restrict__ hyp, const double* __restrict__ jitter, int add_diag,
                                                          double* __restrict__ S, long strideS) {
  const int j = blockIdx.z;
  const int r0 = blockIdx.y * BT_ROWS;
  const int gc = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (blockIdx.x * 512 + 511 < r0) return;               // whole tile strictly below the diagonal
  if (gc >= Np) return;                                  // (Np is even: the pair is inside or outside together)
  const double* __restrict__ X = Xs + (long)j * strideXs;
  const double variance = hyp[j].variance;
  const double dg = add_diag ? (hyp[j].noise + 1e-8 + (jitter ? jitter[j] : 0.0)) : 0.0;
  double xa[D], xb[D];
#pragma unroll
  for (int q = 0; q < D; ++q) {
    xa[q] = gc < N ? X[(long)gc * D + q] : 0.0;
    xb[q] = gc + 1 < N ? X[(long)(gc + 1) * D + q] : 0.0;
  }
  double* __restrict__ Sj = S + (long)j * strideS;
  // two rows per iteration: four independent exp chains per thread (0.122 -> 0.117 ms at config 3)
  for (int rr = 0; rr < BT_ROWS; rr += 2) {
    double v[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int gr = r0 + rr + e;
      if (gr < N) {
        double ra = 0.0, rb = 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) {
          const double xq = X[(long)gr * D + q];
          const double d0 = xq - xa[q], d1 = xq - xb[q];
          ra += d0 * d0;
          rb += d1 * d1;
        }
        // (both kernel values are computed unconditionally and masked afterwards: written as "column valid ? k(r) : 0" the compiler
        // puts every exp behind its own exec-mask branch and the independent chains no longer interleave: 0.134 -> 0.122 ms;
        // tools/kbuild_probe.hip is the same store pattern with and without the arithmetic)
        const double k0 = kern_of_r2(KID, variance, ra), k1 = kern_of_r2(KID, variance, rb);
        v[e][0] = gr == gc ? variance + dg : (gc < N ? k0 : 0.0);          // r = 0 on the diagonal (stationary.py:137, se.py:57-58)
        v[e][1] = gr == gc + 1 ? variance + dg : (gc + 1 < N ? k1 : 0.0);
      } else {
        v[e][0] = (gr == gc) ? 1.0 : 0.0;            // identity padding
        v[e][1] = (gr == gc + 1) ? 1.0 : 0.0;
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) *reinterpret_cast<v2d_f*>(Sj + (long)(r0 + rr + e) * Np + gc) = (v2d_f){v[e][0], v[e][1]};
  }
}

// The same tile with the ROW points staged through LDS (round 3; north_star: "coalesced HBM loads of X into LDS tiles").  What the
// counters said about the kernel above (profiles/r02): 52 % of the wave cycles stalled on instruction issue, no LDS use -- its row
// coordinates are scalar loads INSIDE the "row is real" branch, so every iteration starts with a scalar-memory round trip that nothing
// overlaps.  Here the 64 x D row coordinates of the tile are loaded once, coalesced, into LDS (4 KiB at D = 8); the row loop has no
// branch (padding rows / columns are masked after the arithmetic) and reads its row points as LDS broadcasts (all lanes one address:
// no bank conflict), FOUR rows per iteration = eight independent exp chains per thread, whose loads the compiler issues ahead of
// the arithmetic.  Per element the arithmetic is identical to the kernel above (same differences, same FMA order, same exp): same bits.
template <int D, int KID>
__global__ __launch_bounds__(256) void build_train_lds_kernel(const double* __restrict__ Xs, long strideXs, int N, int Np,
                                                              const KernHyp* __restrict__ hyp, const double* __restrict__ jitter, int add_diag,
                                                              double* __restrict__ S, long strideS) {
  __shared__ double xr[BT_ROWS * D];
  const int j = blockIdx.z;
  const int r0 = blockIdx.y * BT_ROWS;
  const int gc = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (blockIdx.x * 512 + 511 < r0) return;               // whole tile strictly below the diagonal (uniform per workgroup)
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
  dim3 grid((unsigned)((Np + 511) / 512), (unsigned)(Np / BT_ROWS), (unsigned)m);
  const int kid = kernel_id <= 1 ? 0 : kernel_id;
  static const bool scalar_rows = getenv("BOCF_KBUILD_SCALAR") != nullptr;      // A/B: the round-2 kernel (row points by scalar loads)
#define LAUNCH(D, KID)                                                                                                              \
  do {                                                                                                                              \
    if (scalar_rows)                                                                                                                \
      BOCF_LAUNCH((build_train_kernel<D, KID>), grid, dim3(256), 0, s, Xs, strideXs, N, Np, hyp, jitter, add_diag, S, strideS);      \
    else                                                                                                                            \
      BOCF_LAUNCH((build_train_lds_kernel<D, KID>), grid, dim3(256), 0, s, Xs, strideXs, N, Np, hyp, jitter, add_diag, S, strideS);  \
  } while (0)
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

// Cholesky (upper form) of a 128x128 block held in registers, cyclic layout: thread (ty,tx) owns rows ty+16i, columns
// tx+16j.  Blocked by 16 rows: the 16 elimination steps of a panel only touch the panel's own rows (one LDS row
// broadcast, one barrier and 8 FMAs per thread each -- this is the serial chain), then the 16 finished rows are applied
// to the trailing tiles (upper block-triangle only) as one barrier-free rank-16 update from the LDS panel `pan`
// (16 x 128).  Every element sees its updates in the same order as an unblocked sweep, so the factor is bit-identical.
// (Measured: no faster than the unblocked sweep -- 95 us per 128-block either way.  The time is the serial chain of 128
// pivots: shuffle, fp64 sqrt + divide, LDS round trip and barrier, ~370 ns each; the FMAs this blocking halves were
// never the bottleneck.  What would shorten it is a wave-local 16x16 diagonal factorization -- no workgroup barrier inside
// the chain -- which this blocking is the scaffold for.)
// A non-positive pivot records *info_j = first_index + k + 1 once and carries on with a unit pivot.
// Only the first `nact` 16-row panels are processed: the caller guarantees that everything beyond them is identity padding
// (rows and columns), which a factorization leaves as it is.
__device__ __forceinline__ void chol128_regs(double (&a)[8][8], double (*pan)[NB], double* invd, int ty, int tx, int* info_j,
                                             int first_index, int nact) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    if (kb >= nact) continue;                            // workgroup-uniform
#pragma unroll 1
    for (int kk = 0; kk < 16; ++kk) {
      const int k = kb * 16 + kk;
      double* rb = pan[kk];
      if (ty == kk) {                                    // the 16 lanes (one quarter-wave) that own row k
        double piv = __shfl(a[kb][kb], (ty & 3) * 16 + kk, 64);
        if (!(piv > 0.0)) {
          if (tx == kk && *info_j == 0) *info_j = first_index + k + 1;
          piv = 1.0;
        }
        const double ukk = sqrt_pos_normal(piv);           // (same bits as sqrt / 1.0 / x in range: see the helpers)
        const double inv = rcp_pos_normal(ukk);
#pragma unroll
        for (int j = kb; j < 8; ++j) {
          const int c = tx + 16 * j;
          const double v = a[kb][j] * inv;
          if (c > k) a[kb][j] = v;
          else if (c == k) a[kb][j] = ukk;
          rb[c] = (c > k) ? v : 0.0;                     // zero for c <= k
        }
        if (tx == kk) invd[k] = inv;
      }
      __syncthreads();
      if (ty > kk) {                                     // the panel's remaining rows (same register row kb)
        const double ur = rb[ty + 16 * kb];
#pragma unroll
        for (int j = kb; j < 8; ++j) a[kb][j] -= ur * rb[tx + 16 * j];
      }
    }
    if (kb < 7 && kb + 1 < nact) {
#pragma unroll 4
      for (int kk = 0; kk < 16; ++kk) {                  // rank-16 update of the trailing tiles i > kb, j >= i
        const double* rb = pan[kk];
        double ur[8], uc[8];
#pragma unroll
        for (int i = kb + 1; i < 8; ++i) {
          ur[i] = rb[ty + 16 * i];
          uc[i] = rb[tx + 16 * i];
        }
#pragma unroll
        for (int i = kb + 1; i < 8; ++i)
#pragma unroll
          for (int j = i; j < 8; ++j) a[i][j] -= ur[i] * uc[j];
      }
      __syncthreads();                                   // the next panel overwrites pan
    }
  }
}

// E = U^-1 of the upper factor whose image is Ul (row stride 129) and whose reciprocal pivots are invd, rows from the
// bottom up: row k = (e_k - sum_{k' > k} U[k][k'] E[k']) / U[k][k].  Same 16-row blocking as chol128_regs: the 16 steps of
// a panel update the panel's own rows, then the finished rows are applied to all rows above as one rank-16 update.
// Needs a barrier between filling Ul and the call.
__device__ __forceinline__ void inv128_regs(double (&e)[8][8], const double* Ul, double (*pan)[NB], const double* invd, int ty, int tx, int nact) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) e[i][j] = (ty + 16 * i == tx + 16 * j) ? 1.0 : 0.0;
#pragma unroll
  for (int kb = 7; kb >= 0; --kb) {
    if (kb >= nact) continue;                            // identity padding: its rows of the inverse stay identity
#pragma unroll 1
    for (int kk = 15; kk >= 0; --kk) {
      const int k = kb * 16 + kk;
      double* rb = pan[kk];
      if (ty == kk) {
        const double inv = invd[k];
#pragma unroll
        for (int j = kb; j < 8; ++j) {
          const int c = tx + 16 * j;
          const double v = (c >= k) ? e[kb][j] * inv : 0.0;
          e[kb][j] = v;
          rb[c] = v;
        }
      }
      __syncthreads();
      if (ty < kk) {                                     // rows above k inside the panel
        const double ur = Ul[(ty + 16 * kb) * 129 + k];
#pragma unroll
        for (int j = kb; j < 8; ++j) e[kb][j] -= ur * rb[tx + 16 * j];
      }
    }
    if (kb > 0) {
#pragma unroll 4
      for (int kk = 15; kk >= 0; --kk) {                 // rank-16 update of the rows of the blocks above
        const int k = kb * 16 + kk;
        const double* rb = pan[kk];
        double ur[8], ec[8];
#pragma unroll
        for (int i = 0; i < kb; ++i) ur[i] = Ul[(ty + 16 * i) * 129 + k];
#pragma unroll
        for (int j = kb; j < 8; ++j) ec[j] = rb[tx + 16 * j];
#pragma unroll
        for (int i = 0; i < kb; ++i)
#pragma unroll
          for (int j = kb; j < 8; ++j) e[i][j] -= ur[i] * ec[j];
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Diagonal block p: A_pp = U^T U (upper, right-looking), then E = U^-1 by back substitution.
// One workgroup per output; the 128x128 block lives in REGISTERS, cyclically distributed
// (thread (ty,tx) owns rows ty+16i, columns tx+16j, i,j < 8), so each of the 128 elimination
// steps costs one 1-KiB LDS broadcast of the pivot row (and pivot column for the inverse), one
// barrier and <= 64 register FMAs per thread.  The block index kb of the step is a compile-time
// loop so every register index is static.
// info[j] = 1-based global index of the first non-positive pivot (LAPACK dpotrf semantics,
// GPy/util/linalg.py:54); the pivot is then replaced by 1 so the remaining arithmetic stays
// finite -- the host restarts with jitter (linalg.py:56-71).
__global__ __launch_bounds__(256, 1) void potrf_diag_kernel(double* __restrict__ S, long strideS, int N, int Np, int p,
                                                            double* __restrict__ E, double* __restrict__ ET, long strideE,
                                                            int* __restrict__ info, int* done) {
  __shared__ double rowbuf[16][NB];                      // the 16 finished rows of the current panel
  __shared__ double invd[NB];
  __shared__ double Ul[NB * 129];                        // U image for the inverse phase (row stride 129: conflict-free column reads)
  const int jo = blockIdx.x;
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = tid & 15;
  // this workgroup is the critical path of the factorization and may share its CU with trailing-update waves of the
  // other stream (lookahead): take instruction-issue priority over them
  __builtin_amdgcn_s_setprio(3);
  double* __restrict__ blk = S + (long)jo * strideS + (long)p * NB * Np + (long)p * NB;
  double a[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a[i][j] = blk[(long)(ty + 16 * i) * Np + tx + 16 * j];

  // ---- Cholesky, upper form
  // panels past the real size of this block are identity padding (the last block of a padded matrix)
  const int nreal = N - p * NB;
  const int nact = nreal >= NB ? 8 : (nreal <= 0 ? 0 : (nreal + 15) >> 4);
  chol128_regs(a, rowbuf, invd, ty, tx, info + jo, p * NB, nact);
  // write U_pp back (upper part; strictly-lower part of the block is zeroed)
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = ty + 16 * i, c = tx + 16 * j;
      const double v = (c >= r) ? a[i][j] : 0.0;
      blk[(long)r * Np + c] = v;
      Ul[r * 129 + c] = v;
    }

  // ---- E = U^-1
  double e[8][8];
  __syncthreads();
  inv128_regs(e, Ul, rowbuf, invd, ty, tx, nact);
  double* __restrict__ Ej = E + (long)jo * strideE + (long)p * NB * NB;
  double* __restrict__ ETj = ET + (long)jo * strideE + (long)p * NB * NB;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = ty + 16 * i, c = tx + 16 * j;
      const double v = (c >= r) ? e[i][j] : 0.0;
      Ej[r * NB + c] = v;
      ETj[c * NB + r] = v;
    }
  dep_signal(done);
}

// ---------------------------------------------------------------------------------------------
// Diagonal block, MFMA form (the default): the same factorization A_pp = U^T U and E = U^-1, organised so that the serial
// part is 8 x (one 16 x 16 factorization inside ONE wave) instead of 2 x 128 pivot steps behind workgroup barriers.
//
//   * The block lives in registers as 16 x 16 tiles in the v_mfma_f64_16x16x4_f64 accumulator layout
//     (lane (c = lane & 15, q = lane >> 4), register r  <->  element [4 r + q][c]).  A tile in that layout is directly
//     the B operand (k = its row index) or, read as the A operand, its transpose -- no data movement.
//   * Forward elimination is applied to the augmented matrix [A | I]:  [A | I] -> [U | L^-1], and L^-1 = U^-T = E^T.
//     The inverse therefore needs no second triangular sweep: its tiles ride along in the same rank-16 MFMA updates.
//   * Step kb (8 of them): the wave that owns tile (kb, kb) factors it (16 pivots: IEEE sqrt + divide, v_readlane
//     broadcasts, no LDS round trip, no barrier) together with its 16 identity columns (-> D^-T); then every wave
//     multiplies its tiles of block row kb by D^-T (4 MFMAs per tile) and publishes them in LDS; then every wave applies
//     the rank-16 update to its tiles below (4 MFMAs per tile).  Two workgroup barriers per step, 16 in all.
//   * Tile ownership: column block J (of A and of the augmented part) belongs to wave J & 3, so the work of every step
//     is spread over the four waves.
// Per-element arithmetic differs from the scalar kernel only in the order of the rank-16 sums (MFMA accumulates k in
// groups of four); pivots are still IEEE sqrt and IEEE divide.
// info semantics as potrf_diag_kernel.

__device__ __forceinline__ void chol16_aug(double (*dsc)[DD_LD], double (*gdd)[DD_LD], int lane, int* info_j, int first_index) {
  double col[16];
  const int cc = lane & 15;
  wave_lds_fence();                                      // the caller's lanes have just written the tile image
#pragma unroll
  for (int r = 0; r < 16; ++r) col[r] = lane < 16 ? dsc[r][cc] : (lane < 32 ? (r == cc ? 1.0 : 0.0) : 0.0);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    double piv = readlane_f64(col[k], k);
    if (!(piv > 0.0)) {
      if (lane == 0 && *info_j == 0) *info_j = first_index + k + 1;
      piv = 1.0;
    }
    const double ukk = sqrt_pos_normal(piv);
    const double inv = rcp_pos_normal(ukk);
    // row k of [U | G]: D lanes c > k scale, c == k take the exact root, c < k lie below the diagonal (zero); the
    // augmented lanes scale
    double rk = col[k] * inv;
    if (lane < 16) rk = cc > k ? rk : (cc == k ? ukk : 0.0);
    col[k] = rk;
#pragma unroll
    for (int i = k + 1; i < 16; ++i) {
      const double uki = readlane_f64(rk, i);            // U[k][i] (D lane i, just scaled)
      col[i] -= uki * rk;
    }
  }
  if (lane < 16) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dsc[r][cc] = r <= cc ? col[r] : 0.0;
  } else if (lane < 32) {
#pragma unroll
    for (int r = 0; r < 16; ++r) gdd[r][cc] = col[r];
  }
  wave_lds_fence();                                      // the caller's lanes read both images back
}

__global__ __launch_bounds__(256, 1) void potrf_diag_mfma_kernel(double* __restrict__ S, long strideS, int N, int Np, int p,
                                                                 double* __restrict__ E, double* __restrict__ ET, long strideE,
                                                                 int* __restrict__ info, int* done) {
  __shared__ double pan[2][16][PAN_LD];                  // block row kb of [U | G] (k-major), double-buffered by kb parity
  __shared__ double dsc[16][DD_LD];                      // diagonal tile on its way into / out of the factoring wave
  __shared__ double gdd[16][DD_LD];                      // G_dd = D^-T of the current step
  const int jo = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c15 = lane & 15, q = lane >> 4;
  (void)N;
  double* __restrict__ blk = S + (long)jo * strideS + (long)p * NB * Np + (long)p * NB;
  // slots: t[jj][I] for column block J = w + 4 jj:  I < J -> A(I, J);  I == J -> A(J, J);  I > J -> B(I, J) (augmented part);
  // gd[jj] = B(J, J).  Plain scalars with compile-time indices only (every loop over jj / I / r is unrolled; the runtime
  // step index kb enters through wave-uniform selects), so the 18 tiles stay in registers.
  double t[2][8][4], gd[2][4];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int J = w + 4 * jj;
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) t[jj][I][r] = I <= J ? blk[(long)(16 * I + 4 * r + q) * Np + 16 * J + c15] : 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) gd[jj][r] = (4 * r + q == c15) ? 1.0 : 0.0;
  }

#pragma unroll 1
  for (int kb = 0; kb < 8; ++kb) {
    const int ow = kb & 3, oj = kb >> 2;                 // owner wave / slot of the diagonal tile
    double (*pn)[PAN_LD] = pan[kb & 1];
    // ---- (a) the owner factors [D | I] (chol16_aug, one wave, through the dsc / gdd images)
    if (w == ow) {
      double dg[4] = {0.0, 0.0, 0.0, 0.0}, gg[4];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int I = 0; I < 8; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r) dg[r] = (jj == oj && I == kb) ? t[jj][I][r] : dg[r];
#pragma unroll
      for (int r = 0; r < 4; ++r) dsc[4 * r + q][c15] = dg[r];
      chol16_aug(dsc, gdd, lane, info + jo, p * NB + 16 * kb);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dg[r] = dsc[4 * r + q][c15];                     // U_dd in accumulator layout
        gg[r] = gdd[4 * r + q][c15];                     // B(kb, kb) = G_dd
        pn[4 * r + q][128 + 16 * kb + c15] = gg[r];      // G(kb, kb) joins the published row
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gd[jj][r] = jj == oj ? gg[r] : gd[jj][r];
#pragma unroll
          for (int I = 0; I < 8; ++I) t[jj][I][r] = (jj == oj && I == kb) ? dg[r] : t[jj][I][r];
        }
    }
    __syncthreads();                                     // B1: gdd is there
    // ---- (b) block row kb of the other column blocks: X <- G_dd X, published k-major
    double ga[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) ga[s4] = gdd[c15][4 * s4 + q];           // A operand: A[m][k] = G_dd[m][k]
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int J = w + 4 * jj;
      if (J != kb) {
        double x[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int I = 0; I < 8; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = I == kb ? t[jj][I][r] : x[r];
        v4d_t y = (v4d_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) y = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[s4], x[s4], y, 0, 0, 0);
        const int colbase = (J > kb ? 0 : 128) + 16 * J;                    // U(kb, J) or G(kb, J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pn[4 * r + q][colbase + c15] = y[r];
#pragma unroll
          for (int I = 0; I < 8; ++I) t[jj][I][r] = I == kb ? y[r] : t[jj][I][r];
        }
      }
    }
    __syncthreads();                                     // B2: the whole row kb of [U | G] is published
    // ---- (c) rank-16 update of the tiles below: slot(I, J) -= U(kb, I)^T [U | G](kb, J), I > kb
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int J = w + 4 * jj;
      double fb[4];
      const int bcol = (J > kb ? 0 : 128) + 16 * J;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) fb[s4] = pn[4 * s4 + q][bcol + c15];
#pragma unroll
      for (int I = 1; I < 8; ++I) {
        // A-type slots (I <= J) take the update when I > kb; B-type slots (I > J) when I > kb and J <= kb
        if (I > kb && (I <= J || J <= kb)) {
          v4d_t acc = (v4d_t){t[jj][I][0], t[jj][I][1], t[jj][I][2], t[jj][I][3]};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const double fa = -pn[4 * s4 + q][16 * I + c15];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb[s4], acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) t[jj][I][r] = acc[r];
        }
      }
    }
  }
  // ---- write back: U (upper tiles; the strictly-lower part of the block is zeroed), E^T = G (lower), E = G^T (upper)
  double* __restrict__ Ej = E + (long)jo * strideE + (long)p * NB * NB;
  double* __restrict__ ETj = ET + (long)jo * strideE + (long)p * NB * NB;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int J = w + 4 * jj;
#pragma unroll
    for (int I = 0; I < 8; ++I) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = 4 * r + q;                        // row inside the tile, column inside the tile = c15
        if (I < J) {                                     // A(I, J) = U tile; mirrored positions are zeros
          blk[(long)(16 * I + rl) * Np + 16 * J + c15] = t[jj][I][r];
          blk[(long)(16 * J + rl) * Np + 16 * I + c15] = 0.0;
          ETj[(16 * I + rl) * NB + 16 * J + c15] = 0.0;  // E^T is lower
          Ej[(16 * J + rl) * NB + 16 * I + c15] = 0.0;   // E is upper
        } else if (I == J) {
          blk[(long)(16 * J + rl) * Np + 16 * J + c15] = t[jj][I][r];        // U_dd (zeros below its diagonal)
          ETj[(16 * J + rl) * NB + 16 * J + c15] = gd[jj][r];               // G_dd (zeros above its diagonal)
          Ej[(16 * J + c15) * NB + 16 * J + rl] = gd[jj][r];
        } else {                                         // B(I, J) = G tile, I > J
          ETj[(16 * I + rl) * NB + 16 * J + c15] = t[jj][I][r];
          Ej[(16 * J + c15) * NB + 16 * I + rl] = t[jj][I][r];
        }
      }
    }
  }
  dep_signal(done);
}

// ---------------------------------------------------------------------------------------------
// Diagonal block, MFMA form with a dedicated FACTOR WAVE (the default).  Same mathematics and the same per-element
// arithmetic as potrf_diag_mfma_kernel above; what changes is who waits for whom.  There, each of the 8 steps is
// {one wave factors the 16 x 16 diagonal tile while three wait; all scale the block row; all apply the rank-16 update} in
// sequence: 8 x (3.3 + 0.5 + 3) us.  Here
//   * wave 0 only factors.  Tile (kb, kb) reaches it through an LDS image with every row but the last applied; it applies
//     row kb - 1 itself (4 MFMAs) and starts the 16 pivots at once -- while the eight worker waves (one column block of
//     [A | I] each: half the tiles per wave, and two waves per SIMD, which is what the fp64 matrix pipe needs to run at its full
//     rate) are still applying row kb - 1 to everything else.  The rank-16 updates leave the critical path.
//   * The workgroup has 12 waves, of which waves 4, 8 and 11 exit at once: with waves dealt to the four SIMDs in turn the
//     factor wave has SIMD 0 to itself (its dependent sqrt -> divide chain is not interleaved with other waves' MFMAs).
//   * The 16 x 16 factorization reads the entries of pivot row k from column k of the (symmetric) trailing tile, i.e. from the
//     registers of lane k BEFORE they are scaled: the 15 - k broadcasts no longer wait for the sqrt -> divide of the pivot.
//     uki = (a[i][k] * inv) is the product the other form computed in lane i as (a[k][i] * inv): same operands, same rounding.
//   * Worker code is unrolled over the 8 steps: tile slots are indexed by literals (no select cascades).
// Barriers: one after the prologue, two per step (gdd ready; block row published).
__device__ __forceinline__ void chol16_sym(double (*dsc)[DD_LD], double (*gdd)[DD_LD], int lane, int* info_j, int first_index) {
  double col[16];
  const int cc = lane & 15;
  int fail = 0;
  wave_lds_fence();                                      // the caller's lanes have just written the tile image
#pragma unroll
  for (int r = 0; r < 16; ++r)                           // D lanes: column cc of the tile mirrored from its upper part; then the identity
    col[r] = lane < 16 ? (r <= cc ? dsc[r][cc] : dsc[cc][r]) : (lane < 32 ? (r == cc ? 1.0 : 0.0) : 0.0);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    double piv = readlane_f64(col[k], k);
    double raw[16];
#pragma unroll
    for (int i = k + 1; i < 16; ++i) raw[i] = readlane_f64(col[i], k);      // a[i][k] = a[k][i], not yet scaled
    const bool bad = !(piv > 0.0);                      // (branch-free inside the pivot loop: the failure is recorded after it)
    fail = (bad && fail == 0) ? first_index + k + 1 : fail;
    piv = bad ? 1.0 : piv;
    const double ukk = sqrt_pos_normal(piv);
    const double inv = rcp_pos_normal(ukk);
    double rk = col[k] * inv;
    if (lane < 16) rk = cc > k ? rk : (cc == k ? ukk : 0.0);
    col[k] = rk;
#pragma unroll
    for (int i = k + 1; i < 16; ++i) {
      const double uki = raw[i] * inv;                   // U[k][i]
      col[i] -= uki * rk;
    }
  }
  if (fail && lane == 0 && *info_j == 0) *info_j = fail;
  if (lane < 16) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dsc[r][cc] = r <= cc ? col[r] : 0.0;
  } else if (lane < 32) {
#pragma unroll
    for (int r = 0; r < 16; ++r) gdd[r][cc] = col[r];
  }
  wave_lds_fence();                                      // the caller's lanes read both images back
}


template <int VAR>
__global__ __launch_bounds__(768, 1) void potrf_diag_fw_kernel(double* __restrict__ S, long strideS, int N, int Np, int p,
                                                               double* __restrict__ E, double* __restrict__ ET, long strideE,
                                                               int* __restrict__ info, int* done) {
  __shared__ double pan[2][16][PAN_LD];                  // block row kb of [U | G] (k-major), double-buffered by kb parity
  __shared__ double img[2][16][DD_LD];                   // diagonal tile kb on its way to the factor wave (by parity), U_dd on its way out
  __shared__ double gdd[16][DD_LD];                      // G_dd = D^-T of the current step
  const int jo = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  (void)N;
  if (wv == 4 || wv == 8 || wv == 11) return;            // SIMD 0 belongs to the factor wave (see above); exited waves leave the barriers
  const unsigned long long tl0 = tl_begin();
  double* __restrict__ blk = S + (long)jo * strideS + (long)p * NB * Np + (long)p * NB;
  double* __restrict__ Ej = E + (long)jo * strideE + (long)p * NB * NB;
  double* __restrict__ ETj = ET + (long)jo * strideE + (long)p * NB * NB;
  potrf_fw_body<VAR>(pan, img, gdd, blk, Ej, ETj, Np, p, info + jo, wv, lane);
  dep_signal(done);
  tl_end(1, tl0);
}

// ---------------------------------------------------------------------------------------------
// Device-side dependencies between kernels of DIFFERENT streams (the reserved-CU factorization schedule): a stream event
// wait costs 10-25 us of stream time on this runtime, a flag costs a kernel boundary.
//   producer: every wave drains its stores, workgroup barrier, ONE lane: agent-scope release, drain, relaxed agent-scope add
//             (dep_signal at the end of the producing kernel; or signal_kernel as its own launch behind a kernel that is not
//             instrumented -- the kernel boundary is the release there);
//   consumer: gate_kernel, a single-wave launch IN FRONT of the consuming kernel on its stream, polls the counters (relaxed
//             agent-scope loads, s_sleep between polls, bounded) -- the consuming kernel then starts behind an ordinary kernel
//             boundary, whose acquire makes the released bytes visible to every one of its workgroups.
// Every counter is zeroed by a hipMemsetAsync before the schedule is enqueued; a poll that runs out writes *err and lets the
// stream drain (the host reports it) instead of hanging the GPU.
__global__ void gate_kernel(const int* f0, int n0, const int* f1, int n1, int* err) {
  if (threadIdx.x != 0) return;
  const unsigned long long tl0 = tl_begin();
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();      // 100 MHz
  for (;;) {
    const bool ok0 = !f0 || __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n0;
    const bool ok1 = !f1 || __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n1;
    if (ok0 && ok1) break;
    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > 20000000LL) {  // 0.2 s: something upstream never arrived
      int expected = 0;
      __hip_atomic_compare_exchange_strong(err, &expected, 600000 + n0, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  tl_end(3, tl0);
}

__global__ void signal_kernel(int* f, int add) {
  const unsigned long long tl0 = tl_begin();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(f, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  tl_end(4, tl0);
}

void launch_gate(const int* f0, int n0, const int* f1, int n1, int* err, hipStream_t s) {
  BOCF_LAUNCH(gate_kernel, dim3(1), dim3(64), 0, s, f0, n0, f1, n1, err);
}
void launch_signal(int* f, int add, hipStream_t s) { BOCF_LAUNCH(signal_kernel, dim3(1), dim3(64), 0, s, f, add); }

// ---------------------------------------------------------------------------------------------
// ONE 128 x 128 tile per output with K = 128:  C = beta C + alpha sum_kk A[kk][r] B[kk][c]  -- the two single-tile products
// on the serial chain of the blocked Cholesky (U[p][p+1] = E_p^T A[p][p+1] and A[p+1][p+1] -= U[p][p+1]^T U[p][p+1]).  The
// general GEMM kernel gives such a tile to ONE workgroup (~24 us: 8 k-steps behind a global-load prologue, then the
// epilogue); here the tile is cut into sixteen 32 x 32 pieces, one WAVE each (4 workgroups of 4 waves per output), operands
// straight from global memory / L2 into MFMA fragments (both are k-major: a 16-lane group reads 128 contiguous bytes), no
// LDS, no barrier: every load of a wave is independent of every other, the 128 MFMAs follow.
// (body as a device function: the stand-alone kernel calls it once per workgroup, the persistent chain kernel once per product;
// A, B, C point at the output's matrices, `strip` = which 32-column strip of the tile row this workgroup of four waves owns)
__device__ __forceinline__ void tile128_body(const double* __restrict__ A, long lda, const double* B, long ldb, double* C, long ldc, double alpha,
                                             double beta, int K, int strip) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c15 = lane & 15, q = lane >> 4;
  // a workgroup owns a 32-column strip (its four waves the four 32-row pieces of it): with C aliasing B (in-place row
  // solve) every read of the strip's B columns is over -- barrier below -- before any piece of the strip is overwritten.
  // blockIdx.x counts strips across `ntiles` tiles side by side (B and C advance by 128 columns per tile, A is shared): the
  // whole row solve U[p][p+1 ...] = E_p^T A[p][p+1 ...] is one such launch.
  // MFMA block i of the wave's piece takes rows 2 l + i (lane l), block j the columns 2 l + j: the two values a lane feeds to its two
  // blocks are 16 contiguous bytes of the k-major operand row (half the load instructions), and a lane's results for the two column blocks
  // are adjacent in the tile row (16-B loads and stores of C).  Per element the same products in the same order as the other map.
  const int r0 = w * 32, c0 = strip * 32;
  const double* Aj = A + r0;          // (wave-uniform: the per-lane part is one constant byte offset of the buffer loads)
  const double* Bj = B + c0;          // may alias C (in-place row solve: this piece reads only its own columns)
  double* Cj = C + c0 + 2 * c15;
  const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Aj), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Bj), 0, -1, 0x00020000);
  const unsigned aoff = (unsigned)(((long)q * lda + 2 * c15) * 8), boff = (unsigned)(((long)q * ldb + 2 * c15) * 8);
  const int lda32 = (int)(lda * 32), ldb32 = (int)(ldb * 32);            // 4 k-rows in bytes
  v4d_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (v4d_t){0.0, 0.0, 0.0, 0.0};
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  v2d_t cin[2][4];
  if (beta != 0.0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) cin[i][r] = *reinterpret_cast<const v2d_t*>(Cj + (long)(r0 + 2 * (4 * r + q) + i) * ldc);
  }
  // batches of 8 k4-steps (K is a multiple of 128: an even number of batches), two register sets of 64 VGPRs: the loads of batch h + 1 are
  // in flight under the 32 MFMAs of batch h (two such waves fit a SIMD)
  v2d_t fa0[8], fb0[8], fa1[8], fb1[8];
  auto loadb = [&](v2d_t (&fa)[8], v2d_t (&fb)[8], int bat) {
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      fa[s4] = __builtin_bit_cast(v2d_t, __builtin_amdgcn_raw_buffer_load_b128(resA, aoff, (bat * 8 + s4) * lda32, 0));
      fb[s4] = __builtin_bit_cast(v2d_t, __builtin_amdgcn_raw_buffer_load_b128(resB, boff, (bat * 8 + s4) * ldb32, 0));
    }
  };
  auto mmab = [&](const v2d_t (&fa)[8], const v2d_t (&fb)[8]) {
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s4][i], fb[s4][j], acc[i][j], 0, 0, 0);
  };
  const int nbat = K / 32;
  loadb(fa0, fb0, 0);
#pragma unroll 1
  for (int bat = 0; bat < nbat; bat += 2) {
    loadb(fa1, fb1, bat + 1);
    mmab(fa0, fb0);
    if (bat + 2 < nbat) loadb(fa0, fb0, bat + 2);
    mmab(fa1, fb1);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v2d_t v = (v2d_t){alpha * acc[i][0][r], alpha * acc[i][1][r]};
      if (beta != 0.0) {
        v[0] += beta * cin[i][r][0];
        v[1] += beta * cin[i][r][1];
      }
      *reinterpret_cast<v2d_t*>(Cj + (long)(r0 + 2 * (4 * r + q) + i) * ldc) = v;
    }
}

__global__ __launch_bounds__(256) void tile128_kernel(const double* __restrict__ A, long lda, long strideA, const double* B, long ldb,
                                                      long strideB, double* C, long ldc, long strideC, double alpha, double beta, int K,
                                                      int* done) {
  const unsigned long long tl0 = tl_begin();
  const int jo = blockIdx.y;
  tile128_body(A + (long)jo * strideA, lda, B + (long)jo * strideB, ldb, C + (long)jo * strideC, ldc, alpha, beta, K, (int)blockIdx.x);
  dep_signal(done);
  tl_end(2, tl0);
}

void launch_tile128(const double* A, long lda, long strideA, const double* B, long ldb, long strideB, double* C, long ldc, long strideC,
                    double alpha, double beta, int m, hipStream_t s, int ntiles, int K, int* done) {
  if (ntiles <= 0) return;
  BOCF_LAUNCH(tile128_kernel, dim3(4 * (unsigned)ntiles, (unsigned)m), dim3(256), 0, s, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, alpha,
              beta, K, done);
}

void launch_potrf_diag(double* S, long strideS, int N, int Np, int p, double* E, double* ET, long strideE, int* info, int m, hipStream_t s,
                       int* done) {
  if (g_potrf_scalar == 1)
    BOCF_LAUNCH(potrf_diag_kernel, dim3((unsigned)m), dim3(256), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
  else if (g_potrf_scalar == 2)
    BOCF_LAUNCH(potrf_diag_mfma_kernel, dim3((unsigned)m), dim3(256), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
#ifdef BOCF_PROBES      // timing-only variants (wrong results): tools builds only
  else if (g_potrf_scalar == 11)
    BOCF_LAUNCH(potrf_diag_fw_kernel<1>, dim3((unsigned)m), dim3(768), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
  else if (g_potrf_scalar == 12)
    BOCF_LAUNCH(potrf_diag_fw_kernel<2>, dim3((unsigned)m), dim3(768), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
  else if (g_potrf_scalar == 13)
    BOCF_LAUNCH(potrf_diag_fw_kernel<3>, dim3((unsigned)m), dim3(768), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
  else if (g_potrf_scalar == 14)
    BOCF_LAUNCH(potrf_diag_fw_kernel<4>, dim3((unsigned)m), dim3(768), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
#endif
  else
    BOCF_LAUNCH(potrf_diag_fw_kernel<0>, dim3((unsigned)m), dim3(768), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
}
void set_potrf_scalar(int on) { g_potrf_scalar = on; }

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
                e = exp(-0.5 * r2);
                v = variance * e;
              } else {
                const double rr = sqrt(r2);
                if (KID == 2) {
                  const double s5r = 2.23606797749978969641 * rr;
                  e = exp(-s5r);
                  v = variance * (1.0 + s5r + (5.0 / 3.0) * r2) * e;
                } else {
                  const double s3r = 1.73205080756887729353 * rr;
                  e = exp(-s3r);
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
  } else {
  // ---- K(X,X) + (noise + 1e-8 + jitter) I, identity padding (build_train_kernel)
  double a[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = ty + 16 * i, c = tx + 16 * j;
      double v;
      if (r < N && c < N) {
        double r2 = 0.0;
        for (int q = 0; q < d; ++q) {
          const double df = xs[r * d + q] - xs[c * d + q];
          r2 += df * df;
        }
        v = kern_of_r2(KID, variance, r2);
        if (r == c) v = variance + dg;
      } else {
        v = (r == c) ? 1.0 : 0.0;
      }
      a[i][j] = v;
    }
  // the 16 x 128 panel buffer of the two triangular phases borrows the scaled-input array (rebuilt afterwards)
  double (*pan)[NB] = reinterpret_cast<double (*)[NB]>(xs);
  static_assert(NB * INF_MAX_D == 16 * NB, "panel buffer aliases xs");
  __syncthreads();                                       // every K element has been built from xs
  chol128_regs(a, pan, invd, ty, tx, &info_s, 0, nact);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = ty + 16 * i, c = tx + 16 * j;
      Ul[r * 129 + c] = (c >= r) ? a[i][j] : 0.0;
    }
  double e[8][8];
  __syncthreads();
  inv128_regs(e, Ul, pan, invd, ty, tx, nact);
  __syncthreads();                                       // every read of the U image / panel is done: overwrite with R, xs
  for (int idx = tid; idx < NB * d; idx += 256) {
    const int i = idx / d, q = idx - i * d;
    xs[idx] = i < N ? X[(long)i * d + q] / hj->ls[q] : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = ty + 16 * i, c = tx + 16 * j;
      Ul[r * 129 + c] = (c >= r) ? e[i][j] : 0.0;
    }
  __syncthreads();
  // ---- alpha = R (R^T yc)   (exact_gaussian_inference.py:51)
  ITS(4);
  if (tid < NB) {
    double t = 0.0;
    for (int r = 0; r <= tid; ++r) t += Ul[r * 129 + tid] * ycs[r];
    tv[tid] = t;
  }
  __syncthreads();
  if (tid < NB) {
    double t = 0.0;
    for (int c = tid; c < NB; ++c) t += Ul[tid * 129 + c] * tv[c];
    al[tid] = t;
  }
  __syncthreads();
  ITS(5);
  // ---- Ky^-1 = R R^T on the upper register tiles (i <= j), fused with the hyper-gradient sums (hypgrad_kernel)
  // Kinv[r_i][c_j] = sum_k R[r_i][k] R[c_j][k] as 128 rank-1 updates of the register tiles (R is upper: the terms with
  // k < max(r, c) are zeros, so no masking is needed); only the tiles i <= j are kept
  double kv8[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) kv8[i][j] = 0.0;
  const int nblk = (N + 15) >> 4;                        // register tiles that hold real rows / columns
#pragma unroll 2
  for (int k = 0; k < nact * 16; ++k) {                  // R is block-diagonal with the identity padding: k beyond the real rows adds zeros
    double ur[8], uc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ur[i] = Ul[(ty + 16 * i) * 129 + k];
      uc[i] = Ul[(tx + 16 * i) * 129 + k];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = i; j < 8; ++j)
        if (j < nblk) kv8[i][j] += ur[i] * uc[j];
  }
  ITS(6);
  __syncthreads();                                       // all reads of R are done: park Kinv in its place (each thread
#pragma unroll                                           // reads back only what it wrote)
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = i; j < 8; ++j) Ul[(ty + 16 * i) * 129 + tx + 16 * j] = kv8[i][j];
#pragma unroll
  for (int q = 0; q < INF_MAX_D; ++q) sl[q] = 0.0;
#pragma unroll 1
  for (int i = 0; i < nblk; ++i) {
#pragma unroll 1
    for (int j = i; j < nblk; ++j) {
      const int r = ty + 16 * i, c = tx + 16 * j;
      if (r >= N || c >= N || c < r) continue;           // upper triangle of the real block; (r,c) stands for (c,r) too
      const double kinv = Ul[r * 129 + c];
      const double g = 0.5 * (al[r] * al[c] - kinv);
      if (r == c) {
        sn += g;
        sv += g;                                         // K_ii / variance = 1
      } else {
        double r2 = 0.0;
        for (int q = 0; q < d; ++q) {
          const double df = xs[r * d + q] - xs[c * d + q];
          r2 += df * df;
        }
        double kv, f;
        if (KID <= 1) {
          kv = variance * exp(-0.5 * r2);
          f = kv;
        } else {
          const double rr = sqrt(r2);
          if (KID == 2) {
            const double s5r = 2.23606797749978969641 * rr, ex = exp(-s5r);
            kv = variance * (1.0 + s5r + (5.0 / 3.0) * r2) * ex;
            f = (5.0 / 3.0) * variance * (1.0 + s5r) * ex;
          } else {
            const double s3r = 1.73205080756887729353 * rr, ex = exp(-s3r);
            kv = variance * (1.0 + s3r) * ex;
            f = 3.0 * variance * ex;
          }
        }
        const double g2 = 2.0 * g;
        sv += g2 * kv / variance;
        const double gf = g2 * f;
#pragma unroll
        for (int q = 0; q < INF_MAX_D; ++q)
          if (q < d) {
            const double df = xs[r * d + q] - xs[c * d + q];
            sl[q] += gf * (df * df);
          }
      }
    }
  }
  }                                                        // (scalar form)
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
  // the matrix-pipe form of the inference at every size (round 3, per leapfrog step of the resident chain, N = 16 / 32 / 64 / 128: 0.032 / 0.040 /
  // 0.052 / 0.092 ms against 0.045 / 0.062 / 0.123 / 0.205 ms for the scalar register-blocked form, which BOCF_INFER_SCALAR selects for A/B)
  const bool mfma = getenv("BOCF_INFER_SCALAR") == nullptr;
#define LAUNCH(KID)                                                                             \
  do {                                                                                          \
    if (mfma) BOCF_LAUNCH((hmc128_kernel<KID, 1>), dim3((unsigned)m), dim3(256), 0, s, a);       \
    else BOCF_LAUNCH((hmc128_kernel<KID, 0>), dim3((unsigned)m), dim3(256), 0, s, a);            \
  } while (0)
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
  const bool scalar = getenv("BOCF_INFER_SCALAR") != nullptr;    // (A/B only, as in launch_hmc128)
#define LAUNCH(KID)                                                                                                \
  do {                                                                                                             \
    if (scalar) BOCF_LAUNCH((infer128_kernel<KID, 0>), dim3((unsigned)m), dim3(256), 0, s, X, N, d, hyp, yc, out);  \
    else BOCF_LAUNCH((infer128_kernel<KID, 1>), dim3((unsigned)m), dim3(256), 0, s, X, N, d, hyp, yc, out);         \
  } while (0)
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
                                                        const double* __restrict__ jitter, const double* __restrict__ alpha, double* __restrict__ part) {
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
    const int kend = (blk + 1) * NB < N ? (blk + 1) * NB : N;
    for (int k = blk * NB; k < kend; ++k) {
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

void launch_kalpha_dd(const double* Xs, long strideXs, int N, int Np, int d, int kernel_id, const KernHyp* hyp, const double* jitter,
                      const double* alpha, double* part, int m, hipStream_t s, const int* kids) {
  if (kids) {
    bocf_family_runs(kernel_id, kids, m, [&](int j0, int mr, int kid_) {
      launch_kalpha_dd(Xs + (long)j0 * strideXs, strideXs, N, Np, d, kid_, hyp + j0, jitter ? jitter + j0 : nullptr, alpha + (long)j0 * Np,
                       part + (long)j0 * (Np / NB) * 2 * Np, mr, s, nullptr);
    });
    return;
  }
  dim3 grid((unsigned)((Np + 255) / 256), (unsigned)(Np / NB), (unsigned)m);
  const int kid = kernel_id <= 1 ? 0 : kernel_id;
#define LAUNCH(D, KID) BOCF_LAUNCH((kalpha_dd_kernel<D, KID>), grid, dim3(256), 0, s, Xs, strideXs, N, Np, hyp, jitter, alpha, part)
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

__global__ void refine_rhs_kernel(const double* __restrict__ part, int N, int Np, const double* __restrict__ yc, double* __restrict__ r) {
  const int j = blockIdx.y, nblk = Np / NB;
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
  BOCF_LAUNCH(refine_rhs_kernel, dim3((unsigned)((Np + 255) / 256), (unsigned)m), dim3(256), 0, s, part, N, Np, yc, r);
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
  if (kid <= 1) return variance * exp(-0.5 * r2);
  const double r = sqrt(r2);
  if (kid == 2) {
    const double s5r = 2.23606797749978969641 * r;
    return (5.0 / 3.0) * variance * (1.0 + s5r) * exp(-s5r);
  }
  return 3.0 * variance * exp(-1.73205080756887729353 * r);
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
          kv = variance * exp(-0.5 * r2);
          f = kv;
        } else {
          const double r = sqrt(r2);
          if (KID == 2) {
            const double s5r = 2.23606797749978969641 * r, e = exp(-s5r);
            kv = variance * (1.0 + s5r + (5.0 / 3.0) * r2) * e;
            f = (5.0 / 3.0) * variance * (1.0 + s5r) * e;
          } else {
            const double s3r = 1.73205080756887729353 * r, e = exp(-s3r);
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
