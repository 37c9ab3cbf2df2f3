// The launched chain kernels of the blocked Cholesky (GPy/util/linalg.py:52-77 dpotrf, :189-223 dtrtri on a 128 x 128 block): the
// diagonal-block kernel with a dedicated factor wave (body: fit_device.h), the wave-level single-tile products of the row solves and thin
// updates, and the gate / signal kernels of the reserved-CU schedule.  Schedules: capi_chol.hip.
#include "bocf_internal.h"
#include "fit_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

static thread_local int g_potrf_scalar = 0;     // probes build only (option "potrf_scalar"): 11..14 = timing-only variants of the diagonal-block kernel; set by the launching thread right before its launches

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

#ifdef BOCF_PROBES
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
#else
__device__ __forceinline__ unsigned long long tl_begin() { return 0ull; }
__device__ __forceinline__ void tl_end(int, unsigned long long) {}
#endif

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
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;      // the attempt is already condemned: ONE time-out, not one per gate
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
#ifdef BOCF_PROBES      // timing-only variants (wrong results): tools builds only
  if (g_potrf_scalar == 11)
    BOCF_LAUNCH(potrf_diag_fw_kernel<1>, dim3((unsigned)m), dim3(768), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
  else if (g_potrf_scalar == 12)
    BOCF_LAUNCH(potrf_diag_fw_kernel<2>, dim3((unsigned)m), dim3(768), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
  else if (g_potrf_scalar == 13)
    BOCF_LAUNCH(potrf_diag_fw_kernel<3>, dim3((unsigned)m), dim3(768), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
  else if (g_potrf_scalar == 14)
    BOCF_LAUNCH(potrf_diag_fw_kernel<4>, dim3((unsigned)m), dim3(768), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
  else
#endif
    BOCF_LAUNCH(potrf_diag_fw_kernel<0>, dim3((unsigned)m), dim3(768), 0, s, S, strideS, N, Np, p, E, ET, strideE, info, done);
}
void set_potrf_scalar(int on) { g_potrf_scalar = on; }   // (read only by the probes build)

