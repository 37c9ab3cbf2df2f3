// Device functions shared by the fit-side translation units (fit.hip, chol_team.hip): the covariance functions, the pivot helpers,
// the 16 x 16 register factorization and the body of the diagonal-block kernel.  gfx950 only.
#pragma once
#include "bocf_internal.h"

#define NB BOCF_TILE

__device__ __forceinline__ double kern_of_r2(int kernel_id, double variance, double r2) {
  // RBF / SE: GPy/kern/src/rbf.py:42-43, se.py:57-60;  Matern52 stationary.py:529-530;  Matern32 :440-441
  if (kernel_id <= 1) return variance * bocf_exp_nonpos(-0.5 * r2);
  const double r = sqrt(r2);
  if (kernel_id == 2) {
    const double s5r = 2.23606797749978969641 * r;   // sqrt(5) r
    return variance * (1.0 + s5r + (5.0 / 3.0) * r2) * bocf_exp_nonpos(-s5r);
  }
  const double s3r = 1.73205080756887729353 * r;
  return variance * (1.0 + s3r) * bocf_exp_nonpos(-s3r);
}

// sqrt of a positive, finite, normal-range double with the instruction sequence the compiler emits for sqrt() -- v_rsq_f64, two
// coupled Newton steps on (g ~ sqrt x, h ~ 1 / (2 sqrt x)), two residual corrections -- minus its range handling (scaling of inputs
// below 2^-767 and the zero / infinity pass-through: four of the ~17 dependent instructions of a pivot's critical path).  Same
// operations on the same operands: the same bits for every input in range; a pivot of a factorization is a kernel variance plus noise
// minus what was eliminated, never a denormal (a non-positive one never gets here).
__device__ __forceinline__ double sqrt_pos_normal(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return g;
}

// 1.0 / d for a positive, finite, normal-range d: v_rcp_f64, two Newton steps, one residual correction -- the compiler's division
// sequence minus v_div_scale / v_div_fixup (which only act outside that range).  tools/sqrt_probe.hip: both helpers agree with sqrt() and
// 1.0 / x bit for bit on 2^32 random doubles in [2^-60, 2^61).
__device__ __forceinline__ double rcp_pos_normal(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  const double e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(e, r, r);
}

typedef double v4d_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

#define PAN_LD 272            // 256 + 16 (2176 B = 128 mod 256 B): panel rows of [U | G], padded so that the four 16-lane k-groups of an 8-byte fragment read hit disjoint banks
#define DD_LD 17

// One wave: Cholesky (upper form) of the 16 x 16 tile image dsc together with 16 identity columns: [D | I] -> [U_dd | G_dd],
// G_dd = U_dd^-T.  Lanes 0..15 hold the columns of D, lanes 16..31 the columns of I (16 registers each); pivots and the row
// entries every lane needs travel through v_readlane (wave-uniform lane index), so a pivot step has no LDS round trip and no
// barrier.  IEEE sqrt and IEEE divide, a[i][c] -= u[k][i] u[k][c] in the order of the unblocked algorithm.  On return dsc
// holds U_dd (zeros below the diagonal) and gdd holds G_dd; both were written by this wave (LDS operations of one wave
// complete in order).  A non-positive pivot records *info_j = first_index + k + 1 once and carries on with a unit pivot.
// LDS hand-off between the lanes of ONE wave: the hardware completes a wave's LDS operations in order, but the compiler
// reasons per thread -- without a fence it may forward a lane's own earlier store to its later load of a word that another
// lane has rewritten in between.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The same factorization of [D | I] -> [U_dd | G_dd] with both tiles IN REGISTERS in the accumulator layout of v_mfma_f64_16x16x4 (lane
// 16 q + c holds rows 4 r + q, column c in element r) and the trailing updates on the matrix pipe: the tile is worked off in four strips
// of four rows; inside a strip the four pivots touch only the strip's own rows (a pivot needs one row broadcast per tile -- ds_bpermute
// from the 16 lanes that hold row p -- and at most three multipliers by v_readlane, against fifteen in chol16_aug), and everything below
// the strip receives the strip as ONE rank-4 update per tile, D -= S^T S and G -= S^T T: a strip in accumulator layout is at once the
// B operand (k = q, n = c) and the A operand of its transpose (m = c, k = q), so the update is mfma(-S, S, D) on registers that are
// already there.  The update also runs over the strip's own (final) rows and the rows above; the latter see exact zeros (U is upper
// triangular), the former are put back.  No LDS, no barrier.  IEEE sqrt and divide as in chol16_aug; a non-positive pivot records
// *info_j = first_index + k + 1 once and carries on with a unit pivot.
__device__ __forceinline__ void chol16_regs(v4d_t& D, v4d_t& G, int lane, int* info_j, int first_index) {
  const int c = lane & 15, q = lane >> 4;
  int fail = 0;
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    double sd = D[kq], sg = G[kq];                       // row 4 kq + q of [D | G], column c
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k = 4 * kq + p;
      double piv = readlane_f64(sd, 16 * p + k);
      const bool bad = !(piv > 0.0);
      fail = (bad && fail == 0) ? first_index + k + 1 : fail;
      piv = bad ? 1.0 : piv;
      const double ukk = sqrt_pos_normal(piv);
      const double inv = rcp_pos_normal(ukk);
      if (q == p) {                                      // row k of [U | G]: scaled; the exact root on the diagonal, zeros below it
        sd = c > k ? sd * inv : (c == k ? ukk : 0.0);
        sg = sg * inv;
      }
      if (p < 3) {
        // rows k + 1 .. 4 kq + 3 of the strip: row_i -= U[k][i] row_k.  U[k][i] comes from lane (p, i), row_k[c] from lane (p, c).
        double mult = 0.0;
#pragma unroll
        for (int q2 = p + 1; q2 < 4; ++q2) {
          const double m = readlane_f64(sd, 16 * p + 4 * kq + q2);
          mult = q == q2 ? m : mult;
        }
        const double rowd = __shfl(sd, 16 * p + c, 64);
        const double rowg = __shfl(sg, 16 * p + c, 64);
        if (q > p) {
          sd = __builtin_fma(-mult, rowd, sd);
          sg = __builtin_fma(-mult, rowg, sg);
        }
      }
    }
    if (kq < 3) {                                        // everything below the strip: one rank-4 update per tile
      D = __builtin_amdgcn_mfma_f64_16x16x4f64(-sd, sd, D, 0, 0, 0);
      G = __builtin_amdgcn_mfma_f64_16x16x4f64(-sd, sg, G, 0, 0, 0);
    }
    D[kq] = sd;
    G[kq] = sg;
  }
  if (fail && lane == 0 && *info_j == 0) *info_j = fail;
}

// VAR != 0: TIMING-ONLY variants (option "potrf_scalar" = 10 + VAR, wrong results): 1 no 16 x 16 factorization, 2 no rank-16 updates, 3 no write-back,
// 4 no block-row products
// The body of the factor-wave kernel as a device function: the stand-alone kernel below calls it once, the persistent chain kernel
// (chol_chain_potrf_kernel) once per panel.  Every live wave of the workgroup must call it (it contains the workgroup barriers: one
// after the prologue, two per step -- the same number on the factor wave's and on the workers' path).
// STREAM (the resident teams, chol_team.hip): every store of the results carries sc1 (written through: other workgroups read them with sc1
// loads, no cache maintenance), and the block rows are PUBLISHED ONE BY ONE -- when the workgroup passes the first barrier of step kb every
// store of step kb - 1 has completed (the workers drain theirs; the factor wave has only this step's twelve in flight) and one lane adds 1 to
// *fstep: row block kb - 1 of [U | G] is in memory.  The last row block is published by the caller's own signal after the call.
typedef unsigned fd_v2u_t __attribute__((ext_vector_type(2)));
// (NOT __builtin_bit_cast(fd_v2u_t, x[r]) on an element of a vector: this compiler then stores element 0 for every r)
__device__ __forceinline__ fd_v2u_t fd_bits(double v) { return (fd_v2u_t){(unsigned)__double2loint(v), (unsigned)__double2hiint(v)}; }
template <bool SC1>
__device__ __forceinline__ void fd_store(double* base, const __amdgpu_buffer_rsrc_t& rs, long idx, double v) {
  if (SC1) __builtin_amdgcn_raw_buffer_store_b64(fd_bits(v), rs, (unsigned)(idx * 8), 0, 16);
  else base[idx] = v;
}
template <int VAR, bool STREAM = false>
__device__ __forceinline__ void potrf_fw_body(double (*pan)[16][PAN_LD], double (*img)[16][DD_LD], double (*gdd)[DD_LD], double* __restrict__ blk,
                                              double* __restrict__ Ej, double* __restrict__ ETj, int Np, int p, int* info_j, int wv, int lane,
                                              int* fstep = nullptr) {
  const int c15 = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(blk, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsET = __builtin_amdgcn_make_buffer_rsrc(ETj, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(Ej, 0, -1, 0x00020000);
  if (wv == 0) {
    // ---------------- factor wave
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();                                     // B0: image of tile (0, 0)
#pragma unroll 1
    for (int kb = 0; kb < 8; ++kb) {
      double (*im)[DD_LD] = img[kb & 1];
      v4d_t d = (v4d_t){im[q][c15], im[4 + q][c15], im[8 + q][c15], im[12 + q][c15]};
      if (kb > 0) {                                      // the last missing row: D -= U(kb-1, kb)^T U(kb-1, kb)
        double (*pm)[PAN_LD] = pan[(kb - 1) & 1];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const double u = pm[4 * s4 + q][16 * kb + c15];
          d = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, u, d, 0, 0, 0);
        }
      }
      // [D | I] -> [U_dd | G_dd] in the tile registers (strips of four rows, rank-4 updates on the matrix pipe: chol16_regs)
      v4d_t g = (v4d_t){q == c15 ? 1.0 : 0.0, 4 + q == c15 ? 1.0 : 0.0, 8 + q == c15 ? 1.0 : 0.0, 12 + q == c15 ? 1.0 : 0.0};
      if (VAR != 1) chol16_regs(d, g, lane, info_j, p * NB + 16 * kb);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = 4 * r + q;
        const double dg = d[r], gg = g[r];
        gdd[rl][c15] = gg;                               // the workers' A operand of the block-row product
        pan[kb & 1][rl][128 + 16 * kb + c15] = gg;       // G(kb, kb) joins the published row
        fd_store<STREAM>(blk, rsB, (long)(16 * kb + rl) * Np + 16 * kb + c15, dg);              // U_dd (zeros below its diagonal)
        fd_store<STREAM>(ETj, rsET, (16 * kb + rl) * NB + 16 * kb + c15, gg);                    // G_dd (zeros above its diagonal)
        fd_store<STREAM>(Ej, rsE, (16 * kb + c15) * NB + 16 * kb + rl, gg);
      }
      if (STREAM) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");       // the stores of step kb - 1 (only this step's twelve may be in flight)
      __syncthreads();                                   // B1: gdd is there
      __syncthreads();                                   // B2: the whole row kb of [U | G] is published
    }
  } else {
    // ---------------- worker: column block J of [A | I].  Slots t[I]: I < J -> A(I, J); I == J -> A(J, J) until it is handed
    // to the factor wave; I > J -> B(I, J) (augmented part, zero until row J is done: B(J, J) = G_dd itself stays with the factor wave)
    const int J = wv < 4 ? wv - 1 : (wv < 8 ? wv - 2 : wv - 3);
    double t[8][4];
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) t[I][r] = I <= J ? blk[(long)(16 * I + 4 * r + q) * Np + 16 * J + c15] : 0.0;
    auto hand_over = [&](int par) {                      // this wave's diagonal tile -> image
#pragma unroll
      for (int I = 0; I < 8; ++I)
        if (I == J) {
#pragma unroll
          for (int r = 0; r < 4; ++r) img[par][4 * r + q][c15] = t[I][r];
        }
    };
    if (J == 0) hand_over(0);
    // zeros of the mirrored positions (nobody reads them inside this kernel): out of the way before the chain starts
#pragma unroll
    for (int I = 0; I < 8; ++I)
      if (VAR != 3 && I < J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = 4 * r + q;
          fd_store<STREAM>(blk, rsB, (long)(16 * J + rl) * Np + 16 * I + c15, 0.0);                // strictly-lower part of the block
          fd_store<STREAM>(ETj, rsET, (16 * I + rl) * NB + 16 * J + c15, 0.0);                     // E^T is lower
          fd_store<STREAM>(Ej, rsE, (16 * J + rl) * NB + 16 * I + c15, 0.0);                       // E is upper
        }
      }
    __syncthreads();                                     // B0
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      // ---- (c) rank-16 update with row kb - 1: slot(I, J) -= U(kb-1, I)^T [U | G](kb-1, J) for I >= kb; the diagonal slot I == J
      // takes it only while J > kb (row J - 1 is the factor wave's)
      if (kb > 0) {
        double (*pm)[PAN_LD] = pan[(kb - 1) & 1];
        const int bcol = (J > kb - 1 ? 0 : 128) + 16 * J;
        double fb[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) fb[s4] = pm[4 * s4 + q][bcol + c15];
#pragma unroll
        for (int I = 1; I < 8; ++I) {
          if (VAR != 2 && I >= kb && (I < J || (I == J && J > kb) || (I > J && J <= kb - 1))) {
            v4d_t acc = (v4d_t){t[I][0], t[I][1], t[I][2], t[I][3]};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              const double fa = -pm[4 * s4 + q][16 * I + c15];
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb[s4], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) t[I][r] = acc[r];
          }
        }
      }
      if (J == kb + 1) hand_over((kb + 1) & 1);
      if (STREAM && kb > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores of row kb - 1
      __syncthreads();                                   // B1: gdd is there
      if (STREAM && kb > 0 && J == 0 && lane == 0) __hip_atomic_fetch_add(fstep, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // row block kb - 1 is in memory
      // ---- (b) this wave's tile of block row kb: X <- G_dd X, published k-major (the diagonal tile is the factor wave's)
      if (VAR != 4 && J != kb) {
        double ga[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) ga[s4] = gdd[c15][4 * s4 + q];        // A operand: A[m][k] = G_dd[m][k]
        v4d_t y = (v4d_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) y = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[s4], t[kb][s4], y, 0, 0, 0);
        const int colbase = (J > kb ? 0 : 128) + 16 * J;                    // U(kb, J) or G(kb, J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = 4 * r + q;
          pan[kb & 1][rl][colbase + c15] = y[r];
          t[kb][r] = y[r];
          if (VAR != 3) {                                // final: written back here, underneath the following steps
            if (J > kb) {
              fd_store<STREAM>(blk, rsB, (long)(16 * kb + rl) * Np + 16 * J + c15, y[r]);          // U(kb, J)
            } else {
              fd_store<STREAM>(ETj, rsET, (16 * kb + rl) * NB + 16 * J + c15, y[r]);               // G(kb, J): E^T = G (lower), E = G^T (upper)
              fd_store<STREAM>(Ej, rsE, (16 * J + c15) * NB + 16 * kb + rl, y[r]);
            }
          }
        }
      }
      __syncthreads();                                   // B2: the whole row kb of [U | G] is published
    }
  }
}

// paramz 0.9.1 transformations.Logexp (restated in bocf_amd/hyper.py): the constraint of every sampled hyper-parameter
__device__ __forceinline__ double hmc_logexp_f(double x) {           // paramz Logexp.f
  if (x > 36.0) return x;
  const double lim = 709.782712893384;                               // log(DBL_MAX)
  const double c = x < -lim ? -lim : x;
  return log1p(exp(c));
}
__device__ __forceinline__ double hmc_logexp_finv(double f) { return f > 36.0 ? f : log(expm1(f)); }

// Sum over the per-workgroup partials of hypgrad_kernel for component t of output j, by ONE wave in a fixed order (lane l adds blocks l,
// l + 64, ... in turn, then the 64 lane sums are added in lane order): the same bits from hypgrad_reduce_kernel (bocf_lml_gradients) and from
// the POST launch of the stream-resident HMC chain.  Every lane of the wave must call it; the result is valid on every lane.
__device__ __forceinline__ double hypgrad_partial_sum(const double* part, long j, int nblk, int ncomp, int t, int lane) {
  double s = 0.0;
  for (int b = lane; b < nblk; b += 64) s += part[(j * nblk + b) * ncomp + t];
  double tot = 0.0;
  for (int l = 0; l < 64; ++l) tot += __shfl(s, l, 64);
  return tot;
}
