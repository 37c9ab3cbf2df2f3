// One-launch factorization of models with few panels (2 <= nb = Np / 128 <= TEAM_MAX_NB): the blocked Cholesky Ky = U^T U of
// GPy/util/linalg.py:52-77 (jitchol -> dpotrf) and the triangular inverse R = U^-1 of pdinv (linalg.py:189-210, dtrtri) for every output in
// ONE kernel launch.  At these sizes the launched schedule (capi_chol.hip) is a chain of ~10 short dependent launches per panel with the
// chip idle underneath; here each output has a TEAM of workgroups that stay resident for the whole factorization and hand tiles to each
// other through device-scope counters:
//
//   * workgroup 0 of a team factors the diagonal blocks one after the other (the body of potrf_diag_fw_kernel, fit_device.h) and
//     publishes E_p = U_pp^-1;
//   * the other workgroups own UNITS -- halves (128 rows x 64 columns) of the tiles of U and of R^T -- and apply to them, in panel
//     order, the two product forms of the right-looking algorithm, eagerly (as soon as their operands are published):
//       X (solve)    U[p][c]    = E_p^T A[p][c]                               after panel p's diagonal block
//                    R^T[c][r]  = -E_c^T T[c][r]                               (T = the unit's own accumulated sum)
//       Y (update)   A[r][c]   -= U[p][r]^T U[p][c]            r > p          after both operand tiles of row p are solved
//                    T[c][r]   += U[p][c]^T R^T[p][r]          r <= p < c     (R^T[p][p] = E_p^T)
//     Every unit has exactly ONE owner, which is the only workgroup that ever writes it: updates need no atomics and no ordering
//     between workgroups beyond "operand published".  The inverse rides along one panel behind the factorization: R^T[c][r] is
//     finished right after diagonal block c (column-wise forward recurrence R[r][c] = -(sum_{k=r}^{c-1} R[r][k] U[k][c]) E_c), so
//     when the last diagonal block is done one product per unit of the last block row is all that is left.
//   * The units of the diagonal and of the first super-diagonal -- the factorization's critical chain
//     potrf(p) -> U[p][p+1] -> A[p+1][p+1] -> potrf(p+1) -- get workgroups of their own when the team is large enough.
//
// Hand-off protocol (MI355X: per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed by other CUs' stores):
//   diagonal blocks (plain loads and stores inside potrf_fw_body):
//     producer: every wave drains its stores, workgroup barrier, ONE lane: agent-scope release, drain, relaxed agent-scope add;
//     consumer: ONE lane polls (relaxed agent-scope loads, bounded), agent-scope acquire, drain, workgroup barrier, plain loads;
//   units (team_tile): EVERY load and store of unit data carries sc1 (loads bypass the L1, stores write through to memory), so the
//     producer only drains (every wave), barrier, ONE lane adds; the consumer polls, barrier, loads -- no cache maintenance on either
//     side (MI355X_MICROARCH.md, "Valid forms": sc1 both sides, one lane signalling behind the barrier).
// Every poll is bounded (0.2 s) and gives up at once when another poll has already timed out: a schedule that cannot complete
// (a workgroup that was never placed) ends with *err != 0 and garbage in the buffers -- the caller redoes the attempt with the
// launched schedule -- instead of hanging the GPU.  All workgroups of the grid must be resident at the same time: the launcher
// sizes the grid to at most one workgroup per compute unit.
#include "fit_device.h"

#define TEAM_SC1 16             // cache policy of every tile load and store: sc1 (device scope: loads bypass the L1, stores write through)
#define TEAM_AGG 4               // beyond 8 panels: rows a unit lets pile up before it takes them in one product
#define TEAM_MAXU 4096           // units one workgroup can own (a team of two workgroups at nb = 32)

typedef double v2d_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ long long team_now() { return (long long)__builtin_amdgcn_s_memrealtime(); }      // 100 MHz
__device__ __forceinline__ int team_ld(const int* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One lane waits until up to three counters have reached their targets, then the whole workgroup acquires.  `id` names the wait:
// the FIRST one that runs out of polls is what *err holds.
template <bool ACQ>
__device__ __forceinline__ void team_wait(const int* f0, int n0, const int* f1, int n1, const int* f2, int n2, int* err, int id) {
  if (threadIdx.x == 0) {
    const long long t0 = team_now();
    // (the counters of one poll are loaded together: one round trip; the error word and the clock only every 16th poll -- a poll is
    // one memory round trip, on the critical chain half of it is waiting time)
    for (unsigned it = 0;; ++it) {
      const int v0 = f0 ? team_ld(f0) : n0, v1 = f1 ? team_ld(f1) : n1, v2 = f2 ? team_ld(f2) : n2;
      if (v0 >= n0 && v1 >= n1 && v2 >= n2) break;
      if ((it & 15u) == 15u) {
        if (team_ld(err) != 0) break;                            // the attempt is already condemned: do not burn another time-out
        if (team_now() - t0 > 20000000LL) {
          int expected = 0;
          __hip_atomic_compare_exchange_strong(err, &expected, id, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (ACQ) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
}

// One lane waits until, for every row q in [q0, q1], the counters fa[q * sa], fa[q * sa + 1] (both halves of an operand tile) and fb[q * sb]
// (the operand half this unit multiplies) have been set
__device__ __forceinline__ void team_wait_rows(const int* fa, int sa, const int* fb, int sb, int q0, int q1, int* err, int id) {
  if (threadIdx.x == 0) {
    const long long t0 = team_now();
    for (int q = q0; q <= q1; ++q) {
      for (unsigned it = 0;; ++it) {
        const int va = team_ld(fa + (long)q * sa), vb = team_ld(fa + (long)q * sa + 1), vc = team_ld(fb + (long)q * sb);
        if (va >= 1 && vb >= 1 && vc >= 1) break;
        if ((it & 15u) == 15u) {
          if (team_ld(err) != 0) { q = q1; break; }
          if (team_now() - t0 > 20000000LL) {
            int expected = 0;
            __hip_atomic_compare_exchange_strong(err, &expected, id, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q = q1;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
}

template <bool REL>
__device__ __forceinline__ void team_signal(int* f) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (REL) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// C[128 x 64] = beta C + alpha A^T B with K = 128 g (A, B k-major; C may alias B: the in-place solve reads only its own columns, and every
// read is over -- barrier -- before any piece is overwritten).  Eight waves, a 32 x 32 piece each, operands straight from L2 into MFMA
// fragments (the register map of tile128_body, fit.hip: block i of a piece takes rows 2 l + i, block j columns 2 l + j, so a lane's two
// values are 16 contiguous bytes of the operand row).  Two register sets of four k4-steps: the loads of the next set are in flight under
// the 16 MFMAs of the current one; C comes in after the loop (the kernel's register budget is 168 with 12-wave workgroups).
__device__ __forceinline__ void team_tile(const double* A, long lda, const double* B, long ldb, double* C, long ldc, double alpha, double beta,
                                          int w8, int lane, int K = NB, double* CT = nullptr, long ldct = 0) {
  const int c15 = lane & 15, q = lane >> 4;
  const int r0 = (w8 & 3) * 32, c0 = (w8 >> 2) * 32;
  const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A + r0), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(B + c0), 0, -1, 0x00020000);
  const unsigned aoff = (unsigned)(((long)q * lda + 2 * c15) * 8), boff = (unsigned)(((long)q * ldb + 2 * c15) * 8);
  const int lda32 = (int)(lda * 32), ldb32 = (int)(ldb * 32);            // 4 k-rows in bytes
  const __amdgpu_buffer_rsrc_t resC = __builtin_amdgcn_make_buffer_rsrc(C + c0, 0, -1, 0x00020000);
  const int ldc8 = (int)(ldc * 8);
  v4d_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (v4d_t){0.0, 0.0, 0.0, 0.0};
  // (this workgroup's own earlier stores -- the previous task on the same unit -- are visible to all its waves)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  v2d_t fa0[4], fb0[4], fa1[4], fb1[4];
  auto loadb = [&](v2d_t (&fa)[4], v2d_t (&fb)[4], int bat) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      fa[s4] = __builtin_bit_cast(v2d_t, __builtin_amdgcn_raw_buffer_load_b128(resA, aoff, (bat * 4 + s4) * lda32, TEAM_SC1));
      fb[s4] = __builtin_bit_cast(v2d_t, __builtin_amdgcn_raw_buffer_load_b128(resB, boff, (bat * 4 + s4) * ldb32, TEAM_SC1));
    }
  };
  auto mmab = [&](const v2d_t (&fa)[4], const v2d_t (&fb)[4]) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s4][i], fb[s4][j], acc[i][j], 0, 0, 0);
  };
  const int nbat = K / 16;                                 // K is a multiple of 128: an even number of batches
  loadb(fa0, fb0, 0);
#pragma unroll 1
  for (int bat = 0; bat < nbat; bat += 2) {
    loadb(fa1, fb1, bat + 1);
    mmab(fa0, fb0);
    if (bat + 2 < nbat) loadb(fa0, fb0, bat + 2);
    mmab(fa1, fb1);
  }
  v2d_t cin[2][4];
  if (beta != 0.0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cin[i][r] = __builtin_bit_cast(v2d_t, __builtin_amdgcn_raw_buffer_load_b128(resC, (unsigned)(16 * c15), (r0 + 2 * (4 * r + q) + i) * ldc8, TEAM_SC1));
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
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), resC, (unsigned)(16 * c15), (r0 + 2 * (4 * r + q) + i) * ldc8, TEAM_SC1);
      if (CT) {                                            // the transpose of the half tile too (read only after the kernel: plain 8-byte stores)
        CT[(long)(c0 + 2 * c15) * ldct + r0 + 2 * (4 * r + q) + i] = v[0];
        CT[(long)(c0 + 2 * c15 + 1) * ldct + r0 + 2 * (4 * r + q) + i] = v[1];
      }
    }
}

// (arguments of a function that is really called arrive in vector registers: what is wave-uniform has to be said so, or every buffer access
// built on it is wrapped in a loop over the distinct values)
template <class T>
__device__ __forceinline__ T* team_uniform(T* ptr) {
  const unsigned long long v = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int team_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// (the LDS images of the diagonal-block body live HERE: handed in from the kernel they would be generic pointers)
__device__ __attribute__((noinline)) void team_potrf_call(double* blk, double* Ej, double* ETj, int Np, int p, int* info_j, int wv, int lane, int* fstep) {
  __shared__ double pan[2][16][PAN_LD];
  __shared__ double img[2][16][DD_LD];
  __shared__ double gdd[16][DD_LD];
  potrf_fw_body<0, true>(pan, img, gdd, team_uniform(blk), team_uniform(Ej), team_uniform(ETj), team_uniform(Np), team_uniform(p), team_uniform(info_j),
                         team_uniform(wv), lane, team_uniform(fstep));
}

// The critical chain of the factorization is  potrf(p) -> U[p][p+1] = E_p^T A[p][p+1] -> A[p+1][p+1] -= U[p][p+1]^T U[p][p+1] -> potrf(p+1).
// ONE workgroup per panel owns the four units involved and runs the two products UNDERNEATH potrf(p), one 16-row block at a time as the
// diagonal workgroup publishes them (potrf_fw_body<., STREAM>): row block kb of U[p][p+1] is  G[kb][0..kb] A[p][p+1][0..kb]  (G = E_p^T is
// lower triangular: the terms left out are exact zeros, so this IS the K = 128 product of team_tile, bit for bit), and it is at once a
// rank-16 term of the update, accumulated in registers in the same order (k ascending) as the K = 128 product.  A[p][p+1] lives in
// registers for the whole panel (wave w: its column block w as B operands), the finished row block goes to U (sc1) and into an LDS image
// the update reads.  When potrf(p) ends only its last row block is left: 32 + 20 matrix instructions per wave instead of two whole unit
// products and a hand-over between them.  Only the 16 x 16 blocks on / above the diagonal of A[p+1][p+1] are formed (all that
// potrf_fw_body reads).
#define XROW_LD 132
__device__ __attribute__((noinline)) void team_crit_stream(double* Sj_, const double* Ep_, int Np_, int p_, const int* fstep_, const int* fPp_, int* err_, int errid_,
                                                            int w8_, int lane, long long* stamp_ = nullptr) {
  __shared__ double xrow[2][16][XROW_LD];
  double* Sj = team_uniform(Sj_);
  const double* Ep = team_uniform(Ep_);
  const int Np = team_uniform(Np_), p = team_uniform(p_), errid = team_uniform(errid_), w8 = team_uniform(w8_);
  const int* fstep = team_uniform(fstep_);
  const int* fPp = team_uniform(fPp_);
  int* err = team_uniform(err_);
#ifdef BOCF_PROBES
  long long* stamp = team_uniform(stamp_);                 // (timeline of the last row block: wait over | row solved | update done)
#endif
  const int c15 = lane & 15, q = lane >> 4;
  double* Apc = Sj + (long)p * NB * Np + (long)(p + 1) * NB;            // tile (p, p + 1): in A, out U
  double* Ann = Sj + (long)(p + 1) * NB * Np + (long)(p + 1) * NB;      // tile (p + 1, p + 1)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(Apc, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc(Ann, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Ep), 0, -1, 0x00020000);
  const int np8 = Np * 8;
  // (this workgroup's own earlier updates of the two tiles: drained by every wave, then visible to all of them)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  double bt[8][4];                                         // B operands: A[p][p+1][16 j + 4 s + q][16 w8 + c15]
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      bt[j][s4] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsA, (unsigned)((q * Np + 16 * w8 + c15) * 8), (16 * j + 4 * s4) * np8, TEAM_SC1));
  // this wave's blocks (I, Jc), I <= Jc, of the next diagonal tile: the 36 of them dealt round the eight waves
  int bi[5], bj[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    int b = w8 + 8 * i, I = 0;
    while (I < 7 && b >= 8 - I) { b -= 8 - I; ++I; }       // (b < 36 only: the fifth slot of waves 4..7 is never used)
    bi[i] = I;
    bj[i] = I + b;
  }
  const int nblk = w8 < 4 ? 5 : 4;
  v4d_t acc[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) acc[i] = (v4d_t){0.0, 0.0, 0.0, 0.0};
  // row block kb of U[p][p+1] (this wave's 16 columns): G[kb][0..kb] A[0..kb], G = E_p^T as A operand (m = c15, k = q): G[16 kb + c15][k] = E_p[k][16 kb + c15]
  // (measured and dropped: skipping the poll of row blocks known to be out, the next block's loads underneath the update, the last block in
  // quarters with the update's tile loaded in between -- 47.8 us per panel against 46.0 for this plain form)
  auto solve_rows = [&](int kb) {
    v4d_t x = (v4d_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      double ga[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * jh + j <= kb) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            ga[j][s4] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsE, (unsigned)((q * NB + 16 * kb + c15) * 8), (16 * (4 * jh + j) + 4 * s4) * NB * 8, TEAM_SC1));
        }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * jh + j <= kb) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) x = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[j][s4], bt[4 * jh + j][s4], x, 0, 0, 0);
        }
    }
    double (*xr)[XROW_LD] = xrow[kb & 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xr[4 * r + q][16 * w8 + c15] = x[r];
      __builtin_amdgcn_raw_buffer_store_b64(fd_bits(x[r]), rsA, (unsigned)((q * Np + 16 * w8 + c15) * 8), (16 * kb + 4 * r) * np8, TEAM_SC1);
    }
  };
  auto update = [&](int kb) {
    double (*xr)[XROW_LD] = xrow[kb & 1];
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (i < nblk) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[4 * s4 + q][16 * bi[i] + c15], xr[4 * s4 + q][16 * bj[i] + c15], acc[i], 0, 0, 0);
      }
  };
#pragma unroll 1
  for (int kb = 0; kb < 7; ++kb) {
    team_wait<false>(fstep, kb + 1, nullptr, 0, nullptr, 0, err, errid);
    solve_rows(kb);
    __syncthreads();                                       // (two images: the writes of step kb + 2 are behind the barrier of step kb + 1)
    update(kb);
  }
  team_wait<false>(fPp, 1, nullptr, 0, nullptr, 0, err, errid);
#ifdef BOCF_PROBES
  if (stamp && threadIdx.x == 0) stamp[0] = team_now();
#endif
  solve_rows(7);
  // the tile the update lands in (A[p][p+1]'s registers are free now): on its way underneath the last rank-16 term
  double cin[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
    if (i < nblk) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cin[i][r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsN, (unsigned)((q * Np + 16 * bj[i] + c15) * 8), (16 * bi[i] + 4 * r) * np8, TEAM_SC1));
    }
#ifdef BOCF_PROBES
  if (stamp && threadIdx.x == 0) stamp[1] = team_now();
#endif
  __syncthreads();
  update(7);
#pragma unroll
  for (int i = 0; i < 5; ++i)
    if (i < nblk) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = -1.0 * acc[i][r];
        v += 1.0 * cin[i][r];
        __builtin_amdgcn_raw_buffer_store_b64(fd_bits(v), rsN, (unsigned)((q * Np + 16 * bj[i] + c15) * 8), (16 * bi[i] + 4 * r) * np8, TEAM_SC1);
      }
    }
#ifdef BOCF_PROBES
  if (stamp && threadIdx.x == 0) stamp[2] = team_now();
#endif
}

#ifdef BOCF_PROBES
#define TEAM_TL_REC 512
#define TEAM_TL(code, ta, tb, tc)                                                                   \
  do {                                                                                              \
    if (a.tl && threadIdx.x == 0 && tl_n < TEAM_TL_REC) {                                           \
      unsigned long long* r_ = a.tl + ((size_t)blockIdx.x * TEAM_TL_REC + tl_n) * 4;                \
      r_[0] = (unsigned long long)(code); r_[1] = (unsigned long long)(ta); r_[2] = (unsigned long long)(tb); r_[3] = (unsigned long long)(tc); \
      ++tl_n;                                                                                       \
    }                                                                                               \
  } while (0)
#define TEAM_NOW() (a.tl ? team_now() : 0ll)
#else
#define TEAM_TL(code, ta, tb, tc) do { } while (0)
#define TEAM_NOW() 0ll
#endif

// counters of one output (ints): P[nb] | D[nb] | TR[nb][nb][2] | IR[nb][nb][2] | ST[nb] (row blocks of diagonal block p in memory) |
// RD[nb] (units of panel p's critical set that carry every row before p)
__host__ __device__ static inline int team_flag_words(int nb) { return ((4 * nb + 4 * nb * nb + 3) / 4) * 4; }

// (no __restrict__ / const on the matrices: other workgroups write them WHILE this one runs)
__global__ __launch_bounds__(768, 1) void chol_team_kernel(TeamArgs a) {
  __shared__ int ulist[TEAM_MAXU];
  __shared__ unsigned char unext[TEAM_MAXU];               // per unit: the next row of the factor (chol, inverse) / of R^T (Ky^-1) it has not received yet
  __shared__ int nlist_s, last_s, poll_s;
  const int T = a.T, nb = a.nb, Np = a.Np;
  const int jo = blockIdx.x / T, w = blockIdx.x % T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* Sj = a.S + (long)jo * a.strideS;
  double* RTj = a.RT + (long)jo * a.strideS;
  double* Ej = a.E + (long)jo * a.strideE;
  double* ETj = a.ET + (long)jo * a.strideE;
  int* F = a.F + (long)jo * a.fstride;
  int* fP = F;
  int* fD = F + nb;
  int* fTR = F + 2 * nb;
  int* fIR = F + 2 * nb + 2 * nb * nb;
  int* fST = F + 2 * nb + 4 * nb * nb;
  int* fRD = fST + nb;
  // the streamed critical chain needs a workgroup of its own next to the diagonal one, and at least one unit owner
  const bool stream_on = a.stream && T >= 3 && a.p1 - a.p0 >= 2;
#ifdef BOCF_PROBES
  int tl_n = 0;
  __shared__ long long stamp_s[4];
#endif
  if (w == 0) {
    // ---------------- the team's diagonal-block workgroup
    if (wv == 4 || wv == 8 || wv == 11) return;          // SIMD 0 belongs to the factor wave (potrf_diag_fw_kernel)
    const int wr = wv < 4 ? wv : (wv < 8 ? wv - 1 : wv - 2);          // rank among the nine live waves
#pragma unroll 1
    for (int p = a.p0; p < a.p1; ++p) {
      const long long t0 = TEAM_NOW();
      if (p > a.p0) team_wait<true>(fD + p, 2, nullptr, 0, nullptr, 0, a.err, 100000 + jo * 100 + p);     // both halves of A[p][p] carry every earlier row
      const long long t1 = TEAM_NOW();
      team_potrf_call(Sj + (long)p * NB * Np + (long)p * NB, Ej + (long)p * NB * NB, ETj + (long)p * NB * NB, Np, p, a.info + jo, wv, lane, fST + p);
      const long long t2 = TEAM_NOW();
      team_signal<false>(fP + p);                        // (every store of the block carried sc1)
      if (a.do_inverse) {                                // R^T[p][p] = E_p^T, R[p][p] = E_p (read only after the kernel: no hand-off)
        const v2d_t* src = reinterpret_cast<const v2d_t*>(ETj + (long)p * NB * NB);
        const v2d_t* srcE = reinterpret_cast<const v2d_t*>(Ej + (long)p * NB * NB);
        double* Rj = a.R + (long)jo * a.strideS;
        for (int e = wr * 64 + lane; e < NB * NB / 2; e += 9 * 64) {
          const int row = e >> 6, c2 = e & 63;
          *reinterpret_cast<v2d_t*>(RTj + (long)(p * NB + row) * Np + (long)p * NB + 2 * c2) = src[e];
          *reinterpret_cast<v2d_t*>(Rj + (long)(p * NB + row) * Np + (long)p * NB + 2 * c2) = srcE[e];
        }
      }
      TEAM_TL(1000000 + p * 10000, t0, t1, t2);
    }
    return;
  }
  if (wv >= 8) return;
  if (stream_on && w == 1) {
    // ---------------- the team's streaming workgroup: for every panel, U[p][p+1] and the last row of A[p+1][p+1] underneath potrf(p)
#pragma unroll 1
    for (int p = a.p0; p < a.p1 - 1; ++p) {
      const long long t0 = TEAM_NOW();
      if (p > a.p0) team_wait<false>(fRD + p, 4, nullptr, 0, nullptr, 0, a.err, 700000 + jo * 100 + p);   // the four units carry every row before p
      const long long t1 = TEAM_NOW();
#ifdef BOCF_PROBES
      team_crit_stream(Sj, Ej + (long)p * NB * NB, Np, p, fST + p, fP + p, a.err, 710000 + jo * 100 + p, wv, lane, a.tl ? stamp_s : nullptr);
#else
      team_crit_stream(Sj, Ej + (long)p * NB * NB, Np, p, fST + p, fP + p, a.err, 710000 + jo * 100 + p, wv, lane);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(fD + p + 1, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(fTR + (p * nb + p + 1) * 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(fTR + (p * nb + p + 1) * 2 + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      TEAM_TL(7000000 + p * 10000, t0, t1, TEAM_NOW());
#ifdef BOCF_PROBES
      if (a.tl) TEAM_TL(8000000 + p * 10000, stamp_s[0], stamp_s[1], stamp_s[2]);
#endif
    }
    return;
  }
  // ---------------- unit owners
  const int v = stream_on ? w - 2 : w - 1, TW = stream_on ? T - 2 : T - 1;
  if (tid == 0) {
    // canonical enumeration of the team's units; critical ones (diagonal, first super-diagonal) first come first served to workgroups of
    // their own when the team has more workgroups than critical units
    const int p0 = a.p0, p1 = a.p1;
    const int nA = 4 * (p1 - p0 - 1);
    // critical units get workgroups of their own only while the rest of the team can carry the other units at <= a.crit_load apiece (otherwise
    // the few workgroups left for everything else would set the pace: 16 panels on 63 workgroups ran 5x slower that way); else round-robin
    int nbulk = 0;
    for (int r = p0; r < p1; ++r) nbulk += 2 * (nb - r);
    nbulk -= 2 + nA;
    if (a.do_inverse) nbulk += (nb - p0) * (nb - p0 - 1);
    if (a.do_kinv) nbulk += nb * (nb + 1);
    const bool dedicated = TW > nA && nbulk <= a.crit_load * (TW - nA);
    int iA = 0, iB = 0, n = 0, last = p0;
    for (int r = p0; r < p1; ++r)
      for (int c = r; c < nb; ++c) {
        if (c == p0) continue;                           // A[p0][p0] goes straight to the group's first diagonal block
        for (int h = 0; h < 2; ++h) {
          const bool crit = (c == r) || (c == r + 1 && c < p1);
          const int owner = dedicated ? (crit ? iA++ : nA + (iB++ % (TW - nA))) : (iA++ % TW);
          if (owner == v && n < TEAM_MAXU) {
            unext[n] = (unsigned char)p0;
            ulist[n++] = (r << 16) | (c << 8) | h;
            last = r > last ? r : last;
          }
        }
      }
    if (a.do_inverse)                                      // the inverse of the block [p0, nb) this launch factors: units (c, r), p0 <= r < c
      for (int c = p0 + 1; c < nb; ++c)
        for (int r = p0; r < c; ++r)
          for (int h = 0; h < 2; ++h) {
            const int owner = dedicated ? nA + (iB++ % (TW - nA)) : (iA++ % TW);
            if (owner == v && n < TEAM_MAXU) {
              unext[n] = (unsigned char)r;
              ulist[n++] = (1 << 24) | (r << 16) | (c << 8) | h;
              last = c > last ? c : last;
            }
          }
    if (a.do_kinv)                                         // Ky^-1 = R R^T, upper tiles: unit (r, c) receives R^T[k][r]^T R^T[k][c] for k = c ... nb - 1
      for (int r = 0; r < nb; ++r)
        for (int c = r; c < nb; ++c)
          for (int h = 0; h < 2; ++h) {
            const int owner = dedicated ? nA + (iB++ % (TW - nA)) : (iA++ % TW);
            if (owner == v && n < TEAM_MAXU) {
              unext[n] = (unsigned char)c;
              ulist[n++] = (2 << 24) | (r << 16) | (c << 8) | h;
              last = nb - 1;
            }
          }
    nlist_s = n;
    last_s = last;
  }
  __syncthreads();
  const int nlist = nlist_s, last = last_s;
  const int w8 = wv;
  int cursor = 0;                                          // background work goes round the unit list
#pragma unroll 1
  for (int p = a.p0; p <= last; ++p) {
    const double* Ep = Ej + (long)p * NB * NB;
    const double* ETp = ETj + (long)p * NB * NB;
    double* rowp = Sj + (long)p * NB * Np;               // block row p of U
    bool haveP = false;
    // ---- solves of panel p
#pragma unroll 1
    for (int k = 0; k < nlist; ++k) {
      const int u = ulist[k];
      const int inv = u >> 24, r = (u >> 16) & 255, c = (u >> 8) & 255, h = u & 255;
      const bool mine = inv == 1 ? (c == p) : (inv == 0 && r == p && c > p && !(stream_on && c == p + 1 && c < a.p1));   // (the streamed tile is not solved here)
      if (!mine) continue;
      const long long t0 = TEAM_NOW();
      if (!haveP) {
        team_wait<false>(fP + p, 1, nullptr, 0, nullptr, 0, a.err, 200000 + jo * 100 + p);
        haveP = true;
      }
      const long long t1 = TEAM_NOW();
      if (!inv) {
        double* unit = rowp + (long)c * NB + 64 * h;
        team_tile(Ep, NB, unit, Np, unit, Np, 1.0, 0.0, w8, lane);
        team_signal<false>(fTR + (p * nb + c) * 2 + h);
      } else {
        double* unit = RTj + (long)c * NB * Np + (long)r * NB + 64 * h;
        // (R[r][c] = R^T[c][r]^T goes out with it: rows 64 h ... of tile (r, c) of the upper inverse factor)
        team_tile(Ep, NB, unit, Np, unit, Np, -1.0, 0.0, w8, lane, NB, a.R + (long)jo * a.strideS + ((long)r * NB + 64 * h) * Np + (long)c * NB, Np);
        team_signal<false>(fIR + (c * nb + r) * 2 + h);
      }
      TEAM_TL(2000000 + inv * 1000000 + p * 10000 + r * 100 + c, t0, t1, TEAM_NOW());
    }
    // ---- updates.  One routine applies to unit k every row in [its next row, qhi] in ONE product (blocking until those rows are published).
    // Up to 8 panels every unit takes row p right away (K = 128 each: the arithmetic of the launched schedule, bit for bit).  Beyond that only the
    // units whose own solve is at most two panels away are brought up to date at once; the others are BACKGROUND work, taken round-robin,
    // up to TEAM_AGG rows per product, while this workgroup would otherwise wait for the next diagonal block -- with 16+ panels the updates,
    // not the chain, are most of the work, and a workgroup that did them in step order would keep the chain waiting behind them.
    auto apply_one = [&](int k, int qhi) {
      const int u = ulist[k];
      const int kind = u >> 24, r = (u >> 16) & 255, c = (u >> 8) & 255, h = u & 255;
      const int q0 = unext[k];
      if (q0 > qhi) return;
      const long long t0 = TEAM_NOW();
      if (kind == 0) {
        team_wait_rows(fTR + r * 2, nb * 2, fTR + c * 2 + h, nb * 2, q0, qhi, a.err, 300000 + jo * 100 + qhi);
        const long long t1 = TEAM_NOW();
        const double* rows = Sj + (long)q0 * NB * Np;
        team_tile(rows + (long)r * NB, Np, rows + (long)c * NB + 64 * h, Np, Sj + (long)r * NB * Np + (long)c * NB + 64 * h, Np, -1.0, 1.0, w8, lane,
                  (qhi + 1 - q0) * NB);
        if (tid == 0) unext[k] = (unsigned char)(qhi + 1);
        if (!stream_on) {
          if (r == c && qhi == r - 1) team_signal<false>(fD + r);
        } else if (r == c ? qhi == r - 2 : (c == r + 1 && c < a.p1 && qhi == r - 1)) {
          team_signal<false>(fRD + (r == c ? r - 1 : r));  // this unit carries every row before the panel that streams it
        }
        TEAM_TL(4000000 + qhi * 10000 + r * 100 + c, t0, t1, TEAM_NOW());
      } else if (kind == 1) {
        double* unit = RTj + (long)c * NB * Np + (long)r * NB + 64 * h;
        int q = q0;
        if (q == r) {                                      // first term: R^T[r][r] = E_r^T lives in its own buffer
          const int* f0 = fTR + (r * nb + c) * 2;
          team_wait<false>(f0, 1, f0 + 1, 1, fP + r, 1, a.err, 400000 + jo * 100 + qhi);
          team_tile(Sj + (long)r * NB * Np + (long)c * NB, Np, ETj + (long)r * NB * NB + 64 * h, NB, unit, Np, 1.0, 0.0, w8, lane);
          ++q;
        }
        const long long t1 = TEAM_NOW();
        if (q <= qhi) {
          team_wait_rows(fTR + c * 2, nb * 2, fIR + r * 2 + h, nb * 2, q, qhi, a.err, 400000 + jo * 100 + qhi);
          team_tile(Sj + (long)q * NB * Np + (long)c * NB, Np, RTj + (long)q * NB * Np + (long)r * NB + 64 * h, Np, unit, Np, 1.0, 1.0, w8, lane,
                    (qhi + 1 - q) * NB);
        }
        if (tid == 0) unext[k] = (unsigned char)(qhi + 1);
        TEAM_TL(5000000 + qhi * 10000 + r * 100 + c, t0, t1, TEAM_NOW());
      } else {
        double* unit = a.KI + (long)jo * a.strideS + (long)r * NB * Np + (long)c * NB + 64 * h;
        int q = q0;
        if (q == c) {                                      // first term (k = c): R^T[c][c] = E_c^T from its own buffer
          const int* fr = fIR + (c * nb + r) * 2;
          team_wait<false>(fP + c, 1, r < c ? fr : nullptr, 1, r < c ? fr + 1 : nullptr, 1, a.err, 500000 + jo * 100 + qhi);
          const double* ETc = ETj + (long)c * NB * NB;
          team_tile(r == c ? ETc : RTj + (long)c * NB * Np + (long)r * NB, r == c ? NB : Np, ETc + 64 * h, NB, unit, Np, 1.0, 0.0, w8, lane);
          ++q;
        }
        const long long t1 = TEAM_NOW();
        if (q <= qhi) {
          team_wait_rows(fIR + r * 2, nb * 2, fIR + c * 2 + h, nb * 2, q, qhi, a.err, 500000 + jo * 100 + qhi);
          team_tile(RTj + (long)q * NB * Np + (long)r * NB, Np, RTj + (long)q * NB * Np + (long)c * NB + 64 * h, Np, unit, Np, 1.0, 1.0, w8, lane,
                    (qhi + 1 - q) * NB);
        }
        if (tid == 0) unext[k] = (unsigned char)(qhi + 1);
        TEAM_TL(6000000 + qhi * 10000 + r * 100 + c, t0, t1, TEAM_NOW());
      }
      __syncthreads();                                     // (unext[k] is read again by every thread)
    };
    // the last row a unit may receive: chol (r, c): r - 1; inverse (c, r): c - 1; Ky^-1 (r, c): nb - 1;  first row: p0 / r / c
    auto row_hi = [&](int u) { const int kind = u >> 24, r = (u >> 16) & 255, c = (u >> 8) & 255; return kind == 0 ? (stream_on && r == c ? r - 2 : r - 1) : (kind == 1 ? c - 1 : nb - 1); };
    auto row_lo = [&](int u) { const int kind = u >> 24, r = (u >> 16) & 255, c = (u >> 8) & 255; return kind == 0 ? a.p0 : (kind == 1 ? r : c); };
    const bool eager = nb <= 8;
    const int agg = eager ? 1 : TEAM_AGG;
    // WHICH rows share a product is a property of the unit, not of the moment: its rows are cut into fixed chunks -- the first term of an
    // inverse / Ky^-1 unit alone (another operand buffer), then TEAM_AGG rows at a time, the last two rows one by one (they are taken the
    // moment they appear) -- so the sums are the same whenever and by whichever workgroup they are formed (run to run, and between a
    // replicated and an output-sharded fit, whose teams differ in size)
    auto chunk_end = [&](int u, int q) {
      const int kind = u >> 24, lo = row_lo(u), hi = row_hi(u);
      if (kind != 0 && q == lo) return q;
      if (q >= hi - 1) return q;
      const int base = kind == 0 ? lo : lo + 1;
      int e = base + ((q - base) / agg + 1) * agg - 1;
      return e > hi - 2 ? hi - 2 : e;
    };
    auto apply = [&](int k, int qhi) {                     // every whole chunk that ends at or before qhi
      const int u = ulist[k];
      for (;;) {
        const int q0 = unext[k];
        if (q0 > qhi) break;
        const int e = chunk_end(u, q0);
        if (e > qhi) break;
        apply_one(k, e);
      }
    };
#pragma unroll 1
    for (int k = 0; k < nlist; ++k) {
      const int u = ulist[k];
      const int hi = row_hi(u);
      if (row_lo(u) > p || hi < p) continue;               // row p is not one of this unit's rows
      if (eager || hi - p <= 1 || p == nb - 1) apply(k, p);  // due within two panels (or the factorization is over): up to date now
    }
    if (!eager && p < nb - 1) {
      // background: until the next diagonal block is published (or nothing is left that could be done)
      int idle = 0;
      while (idle < nlist) {
        if (tid == 0) poll_s = team_ld(fP + p + 1) >= 1 || team_ld(a.err) != 0 ? 1 : 0;
        __syncthreads();
        const int stop = poll_s;
        __syncthreads();
        if (stop) break;
        const int k = cursor;
        cursor = cursor + 1 == nlist ? 0 : cursor + 1;
        const int u = ulist[k];
        const int q0 = unext[k];
        if (row_lo(u) > p || q0 > row_hi(u) || q0 > p) { ++idle; continue; }
        const int qhi = chunk_end(u, q0);                    // the unit's next chunk: all of it must have been solved
        if (qhi > p) { ++idle; continue; }
        // is the unit's next row published?  (one look, no waiting; the rows behind it are waited for inside apply)
        if (tid == 0) {
          const int kind = u >> 24, r = (u >> 16) & 255, c = (u >> 8) & 255, h = u & 255;
          int ok;
          if (kind == 0) ok = team_ld(fTR + (q0 * nb + r) * 2) >= 1 && team_ld(fTR + (q0 * nb + r) * 2 + 1) >= 1 && team_ld(fTR + (q0 * nb + c) * 2 + h) >= 1;
          else if (kind == 1) ok = q0 == r ? (team_ld(fP + r) >= 1 && team_ld(fTR + (r * nb + c) * 2) >= 1 && team_ld(fTR + (r * nb + c) * 2 + 1) >= 1)
                                           : (team_ld(fTR + (q0 * nb + c) * 2) >= 1 && team_ld(fTR + (q0 * nb + c) * 2 + 1) >= 1 && team_ld(fIR + (q0 * nb + r) * 2 + h) >= 1);
          else ok = q0 == c ? (team_ld(fP + c) >= 1 && (r == c || (team_ld(fIR + (c * nb + r) * 2) >= 1 && team_ld(fIR + (c * nb + r) * 2 + 1) >= 1)))
                            : (team_ld(fIR + (q0 * nb + r) * 2) >= 1 && team_ld(fIR + (q0 * nb + r) * 2 + 1) >= 1 && team_ld(fIR + (q0 * nb + c) * 2 + h) >= 1);
          poll_s = ok;
        }
        __syncthreads();
        const int ready = poll_s;
        __syncthreads();
        if (!ready) { ++idle; continue; }
        apply_one(k, qhi);
        idle = 0;
      }
    }
  }
}

int chol_team_flag_words(int nb) { return team_flag_words(nb); }

// m outputs, T workgroups each; the caller has zeroed the counters and the error word on `s` and guarantees m * T <= compute units
void launch_chol_team(const TeamArgs& a, int m, hipStream_t s) {
  BOCF_LAUNCH(chol_team_kernel, dim3((unsigned)(m * a.T)), dim3(768), 0, s, a);
}
