// fp64 MFMA GEMM for gfx950 (CDNA4):  C[r][c] = beta*Cin + alpha * sum_kk A[kk][r] * B[kk][c].
//
// One kernel serves every dense contraction of the path:
//   * predictive variance  V = R^T K*  (R = U^-1 upper, K* = K(X,X*)), triangular contraction
//     range per row tile, epilogue = column sums of squares (V itself never reaches HBM)
//     -- replaces dtrtrs + square + sum of PosteriorExact.raw_posterior_variance
//        (GPy/inference/latent_function_inference/posterior.py:308-313, GPy/util/linalg.py:91-110)
//   * Cholesky panel solve and symmetric trailing update, triangular inverse
//     -- replaces dpotrf / dtrtri of pdinv (GPy/util/linalg.py:52-71,189-223)
//
// Geometry: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave =
// 4x4 v_mfma_f64_16x16x4_f64 accumulators = 128 VGPRs), BK = 16.  Both operands are k-major,
// so one wave-instruction of a 16-B/lane global load fetches one full 1 KiB k-row of a tile and
// the MFMA fragments are conflict-free ds_read_b64 (row stride 144 doubles: lanes 16..31 of a
// 32-lane LDS group land 32 banks away from lanes 0..15).  Register-staged double buffering:
// tile t+1 is in flight in VGPRs while the 64 MFMAs (4096 cycles/wave) of tile t run.
#include "bocf_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define BM 128
#define BM2 256   // rows of the 256 x 128 tiles
#define BN 128
#define BK 16
#define LDT 144   // padded LDS row (doubles)
#define QCH 16    // rows of the C tile a wave keeps in flight in the store epilogue

// NW = 4 waves (64 x 64 per wave) or 8 waves (64 x 32 per wave, store epilogue, one tile ahead): a wave can issue an fp64 MFMA only
// every second slot of the matrix pipe, so a 4-wave workgroup that is ALONE on its CU (the last, partial round of a launch; every
// workgroup of a launch with fewer workgroups than slots) runs at half rate -- measured on the factorization's trailing updates,
// whose time was ceil(workgroups / 512) x 78 us.  With 8 waves a lone workgroup has the whole pipe and the tail packs by
// throughput.  Same k order per element: identical results.
template <int EPI, int PF, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void gemm_tn_f64_kernel(GemmArgs g) {
  constexpr int JB = NW == 4 ? 4 : 2;                    // 16-column blocks per wave
  constexpr int SP = 16 / NW;                            // staging passes per k-tile (NW k-rows each)
  __shared__ double lds[2][2][BK][LDT];   // [buffer][A|B][k][m or n]   73,728 B

  const int nct = g.Ncols / BN;
  const int nrt = g.M / BM;
  int b = blockIdx.x;
  int rt, ct, batch;
#ifdef BOCF_PROBES        // tile orders that were measured slower (option "swizzle" = 100 + RT, 1): kept for the tools only
  if (g.swizzle >= 100) {
    // Row-group order: RT consecutive row tiles of the SAME column tile are dispatched back to back
    // on the SAME XCD (b % 8 == ct % 8), so they stream one K* column slab through that XCD's L2
    // together (RT-fold fewer HBM reads of K*), while the workgroups in flight still sweep whole
    // K* rows (all column tiles) like the plain order does.  Needs nct % 8 == 0 (launcher checks).
    const int RT = g.swizzle - 100;
    const int per_z = nrt * nct;
    batch = b / per_z;
    const int bb = b - batch * per_z;
    const int grp = bb / (nct * RT);
    const int t = bb - grp * (nct * RT);
    const int ct_hi = t / (8 * RT);
    const int ri = (t >> 3) % RT;
    ct = ct_hi * 8 + (t & 7);
    rt = grp * RT + ri;
  } else if (g.swizzle == 1) {
    // XCD-aware order.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), and
    // an XCD holds 64 of these workgroups at a time: give each XCD one 8x8 super-tile at a time, so
    // the 8 row tiles of a super-tile stream the SAME K* column slabs through that XCD's L2 and the 8
    // column tiles share the same R row panels (speed only -- any placement computes the same values).
    const int nrg = (nrt + 7) >> 3, ncg = (nct + 7) >> 3;
    const int xcd = b & 7, q = b >> 3;
    const int sidx = (q >> 6) * 8 + xcd;
    if (sidx >= nrg * ncg * g.batch) return;
    const int w = q & 63;
    const int cg = sidx % ncg, t = sidx / ncg;
    batch = t % g.batch;
    rt = (t / g.batch) * 8 + (w & 7);
    ct = cg * 8 + (w >> 3);
    if (rt >= nrt || ct >= nct) return;
  } else
#endif
  if (g.swizzle == 2) {
    // row-tile-major across the whole batch: all outputs' heaviest row tiles first (LPT order; the merges of the triangular inverse)
    const int per = nct * g.batch;
    rt = b / per;
    const int rem = b - rt * per;
    batch = rem / nct;
    ct = rem - batch * nct;
  } else if constexpr (EPI == 3) {
    // linear enumeration of the tiles on / above the diagonal (the trailing updates of the factorization: square, uniform K): no workgroup is
    // launched for the strictly lower half (the square grid's early-exit workgroups cost 0.135 ms of the 4.16 ms Cholesky at N = 4096, m = 4)
    const int per = nct * (nct + 1) / 2;
    batch = b / per;
    int ut = b - batch * per;
    rt = 0;
    while (ut >= nct - rt) {                               // row rt holds nct - rt tiles
      ut -= nct - rt;
      ++rt;
    }
    ct = rt + ut;
  } else {
    rt = b / nct;
    ct = b - rt * nct;
    batch = blockIdx.z;
  }
  if (g.rt_desc) rt = nrt - 1 - rt;
  if (g.ct_desc) ct = nct - 1 - ct;
  if (g.upper_only && ct < rt) return;
  int kend = g.kb + g.krt * rt + g.kct * ct;
  if (kend > g.K) kend = g.K;
  const int kbeg = g.kbeg_rt * rt + g.kbeg_ct * ct;
  long offA = (long)batch * g.strideA, offB = (long)batch * g.strideB, offC = (long)batch * g.strideC;
  if (g.batch1 > 0) {                    // two-level batch: z = z2 * batch1 + z1
    const int z1 = batch % g.batch1, z2 = batch / g.batch1;
    offA = (long)z1 * g.strideA + (long)z2 * g.strideA2;
    offB = (long)z1 * g.strideB + (long)z2 * g.strideB2;
    offC = (long)z1 * g.strideC + (long)z2 * g.strideC2;
  }

  const double* __restrict__ A = g.A + offA + (long)rt * BM;
  const double* B = g.B + offB + (long)ct * BN;   // may alias Cout (in-place panel solve)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = NW == 4 ? wave >> 1 : wave >> 2, wc = NW == 4 ? wave & 1 : wave & 3;
  const int l15 = lane & 15, lq = lane >> 4;

  // staging map: SP x (16 B of A) + SP x (16 B of B) per thread per k-tile
  const int srow = tid >> 6;          // + NW*i
  const int scol = (tid & 63) * 2;

  v4d acc[4][JB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  // Register staging, TWO k-tiles deep: while tile t is multiplied out of LDS, tile t+1 sits in
  // one register set (written to the other LDS buffer at the end of the step) and tile t+2's
  // global loads are in flight in the second set -- ~2 x 4096 MFMA cycles of cover for an HBM or
  // Infinity-Cache miss.  The loop is unrolled by two so both register sets are statically named.
  v2d ra0[SP], rb0[SP], ra1[SP], rb1[SP];
  // Operand fetch without vector-ALU address arithmetic (integer VALU instructions do not run in the shadow of the fp64 matrix
  // pipe: profiles/r02/mfma_issue_probe.txt): buffer loads whose resource holds the wave-uniform row pointer of the k-tile (scalar
  // unit), ONE constant per-thread byte offset, the four staged rows as scalar offsets.  (Config 2's variance contraction 0.68 ->
  // 0.76 of peak, fit at N = 4096 7.25 -> 7.05 ms.  The pair-contiguous LDS layout of the three-buffer kernel was tried here too:
  // it doubles the number of fetch instructions, and without that kernel's one-per-MFMA interleave it loses what it gains.)
  const unsigned aoff = (unsigned)(((long)srow * g.lda + scol) * 8), boff = (unsigned)(((long)srow * g.ldb + scol) * 8);
  const int lda32 = (int)(g.lda * 8 * NW), ldb32 = (int)(g.ldb * 8 * NW);  // NW rows in bytes
  auto gload = [&](v2d (&ra)[SP], v2d (&rb)[SP], int kt) {
    const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A + (long)kt * g.lda), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(B + (long)kt * g.ldb), 0, -1, 0x00020000);
#pragma unroll
    for (int i = 0; i < SP; ++i) {
      ra[i] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(resA, aoff, i * lda32, 0));
      rb[i] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(resB, boff, i * ldb32, 0));
    }
  };
  auto lstore = [&](const v2d (&ra)[SP], const v2d (&rb)[SP], int buf) {
#pragma unroll
    for (int i = 0; i < SP; ++i) {
      *reinterpret_cast<v2d*>(&lds[buf][0][srow + NW * i][scol]) = ra[i];
      *reinterpret_cast<v2d*>(&lds[buf][1][srow + NW * i][scol]) = rb[i];
    }
  };
  auto compute = [&](int cur) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kq = ks * 4 + lq;
      double fa[4], fb[JB];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = lds[cur][0][kq][wr * 64 + i * 16 + l15];
#pragma unroll
      for (int j = 0; j < JB; ++j) fb[j] = lds[cur][1][kq][wc * (16 * JB) + j * 16 + l15];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  if constexpr (PF == 2) {
    if (kend > kbeg) {
      gload(ra0, rb0, kbeg);
      if (kbeg + BK < kend) gload(ra1, rb1, kbeg + BK);
      lstore(ra0, rb0, 0);
    }
    __syncthreads();
    for (int kt = kbeg; kt < kend; kt += 2 * BK) {
      // even half: tile kt in LDS buffer 0, tile kt+BK in register set 1
      if (kt + 2 * BK < kend) gload(ra0, rb0, kt + 2 * BK);
      compute(0);
      if (kt + BK < kend) lstore(ra1, rb1, 1);
      __syncthreads();
      if (kt + BK >= kend) break;
      // odd half: tile kt+BK in LDS buffer 1, tile kt+2BK in register set 0
      if (kt + 3 * BK < kend) gload(ra1, rb1, kt + 3 * BK);
      compute(1);
      if (kt + 2 * BK < kend) lstore(ra0, rb0, 0);
      __syncthreads();
    }
  } else {
    // one tile ahead (fewer registers; used by the store-C variant of the fit, which is latency-bound anyway)
    if (kend > kbeg) {
      gload(ra0, rb0, kbeg);
      lstore(ra0, rb0, 0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kbeg; kt < kend; kt += BK) {
      const bool more = (kt + BK) < kend;
      if (more) gload(ra0, rb0, kt + BK);
      compute(cur);
      if (more) lstore(ra0, rb0, cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // accumulator layout of v_mfma_f64_16x16x4_f64: lane holds D[row = (lane>>4) + 4*reg][col = lane&15]
  if (EPI == 0 || EPI == 3) {
    // Store / read-modify-write of the C tile in whole 1-KiB rows: the accumulators (16 columns x 4 rows per instruction,
    // rows 32 KiB apart in memory) are transposed through the now idle operand LDS, 64 tile rows at a time, so that every
    // global access of the epilogue is one wave-wide 16-B-per-lane instruction over ONE contiguous row of the tile.  The
    // short-K updates of the factorization (K = 128: 8 k-steps between a tile read and a tile write) are bound by exactly
    // this traffic.  Same arithmetic per element as the direct form: alpha * acc (+ beta * Cin).
    const double* Cin = g.Cin ? g.Cin + offC : nullptr;
    double* Cout = g.Cout + offC;
    const double alpha = g.alpha, beta = g.beta;
    double* stage = &lds[0][0][0][0];                       // 64 x 128 doubles = 64 KiB of the 72 KiB
    __syncthreads();                                        // every wave is done with the operand tiles
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (wr == half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(i * 16 + lq + 4 * r) * BN + wc * (16 * JB) + j * 16 + l15] = acc[i][j][r];
      }
      __syncthreads();
      const long row0 = (long)rt * BM + half * 64;
      const long col = (long)ct * BN + lane * 2;
      constexpr int RW = 64 / NW, QC = RW < QCH ? RW : QCH;   // rows per wave of a 64-row pass; QC of them in flight (the C reads are latency-bound)
#pragma unroll
      for (int q0 = 0; q0 < RW; q0 += QC) {
        v2d cin[QC];
        if (Cin) {
#pragma unroll
          for (int q = 0; q < QC; ++q) cin[q] = *reinterpret_cast<const v2d*>(Cin + (row0 + wave + NW * (q0 + q)) * g.ldc + col);
        }
#pragma unroll
        for (int q = 0; q < QC; ++q) {
          const int rr = wave + NW * (q0 + q);
          v2d v = *reinterpret_cast<const v2d*>(&stage[rr * BN + lane * 2]);
          v[0] = alpha * v[0];
          v[1] = alpha * v[1];
          if (Cin) {
            v[0] += beta * cin[q][0];
            v[1] += beta * cin[q][1];
          }
          *reinterpret_cast<v2d*>(Cout + (row0 + rr) * g.ldc + col) = v;
        }
      }
      __syncthreads();
    }
  } else {
    // column sums of squares over the tile's 128 rows, fixed summation order (deterministic)
    __syncthreads();
    double* red = &lds[0][0][0][0];   // [2 (wr)][128 cols]
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s = __builtin_fma(acc[i][j][r], acc[i][j][r], s);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lq == 0) red[wr * 128 + wc * (16 * JB) + j * 16 + l15] = s;
    }
    __syncthreads();
    if (tid < 128) {
      double* out = g.sumsq + (long)batch * g.strideSumsq + (long)rt * g.Ncols + (long)ct * BN;
      out[tid] = red[tid] + red[128 + tid];
    }
  }
}


// ---------------------------------------------------------------------------------------------
#ifdef BOCF_PROBES        // the two-buffer 256-row kernel of round 2a (option "swizzle" = 256 / 257, and the carrier of the kstar_valu_probe timing variants)
// 256 x 128 variant of the sum-of-squares contraction (option "swizzle" = 256): one 512-thread workgroup (8 waves, 4 x 2,
// 64 x 64 per wave) computes TWO vertically adjacent 128-row tiles against the same B (K*) column slab, so every K*
// element is fetched once per 256 rows instead of once per 128 -- half the HBM / fabric traffic of the operand that is
// re-read.  Each 128-row half keeps its own contraction length (the upper half of a triangular A stops 128 earlier: its
// waves sit out the last 8 k-steps) and its own partial row of sums, reduced in the same order as the 128 x 128 kernel,
// so the results are bit-identical to it.
#define LDA2 272   // 256 + 16: same bank property as LDT
// VPROBE = 1 is a TIMING-ONLY variant (option "kstar_valu_probe", results unchanged): every k-tile additionally executes the
// vector-ALU work that building the K(X, X*) tile inside this kernel would cost -- per thread 4 kernel values of d = 8
// (8 differences, 8 FMAs and one fp64 exp each; the training point wave-uniform through scalar loads, the candidate in
// registers) -- to measure what that work costs next to the MFMA stream before deciding to fuse.
template <int VPROBE, int PA = 2, int PB = 2>   // PA / PB: where waves 0..3 / 4..7 store their share of the next operand tile (0 start, 1 middle, 2 end of the step)
__global__ __launch_bounds__(512, 1) void gemm_tn_f64_sumsq256_kernel(GemmArgs g) {
  __shared__ double ldsA[2][BK][LDA2];    // 69,632 B
  __shared__ double ldsB[2][BK][LDT];     // 36,864 B
  const int nct = g.Ncols / BN;
  const int nrt2 = g.M / BM2;
  const int b = blockIdx.x;
  int rt2 = b / nct;
  const int ct = b - rt2 * nct;
  const int batch = blockIdx.z;
  if (g.rt_desc) rt2 = nrt2 - 1 - rt2;
  const int rtA = 2 * rt2, rtB = 2 * rt2 + 1;
  int kendA = g.kb + g.krt * rtA + g.kct * ct, kendB = g.kb + g.krt * rtB + g.kct * ct;
  if (kendA > g.K) kendA = g.K;
  if (kendB > g.K) kendB = g.K;
  const int kend = kendA > kendB ? kendA : kendB;
  const long offA = (long)batch * g.strideA, offB = (long)batch * g.strideB;
  const double* __restrict__ A = g.A + offA + (long)rt2 * BM2;
  const double* __restrict__ B = g.B + offB + (long)ct * BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;            // wr 0,1: rows of tile A; 2,3: tile B
  const int l15 = lane & 15, lq = lane >> 4;
  const int my_kend = wr < 2 ? kendA : kendB;
  // staging map per k-tile: A 16 x 256 = 4 x (16 B) per thread, B 16 x 128 = 2 x (16 B) per thread
  const int arow = tid >> 7, acol = (tid & 127) * 2;   // + 4 i
  const int brow = tid >> 6, bcol = (tid & 63) * 2;    // + 8 i

  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  v2d ra0[4], rb0[2], ra1[4], rb1[2];
  auto gload = [&](v2d (&ra)[4], v2d (&rb)[2], int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const v2d*>(A + (long)(kt + arow + 4 * i) * g.lda + acol);
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const v2d*>(B + (long)(kt + brow + 8 * i) * g.ldb + bcol);
  };
  auto lstore = [&](const v2d (&ra)[4], const v2d (&rb)[2], int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<v2d*>(&ldsA[buf][arow + 4 * i][acol]) = ra[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<v2d*>(&ldsB[buf][brow + 8 * i][bcol]) = rb[i];
  };
  auto compute = [&](int cur, int ks0, int ks1) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < ks0 || ks >= ks1) continue;
      const int kq = ks * 4 + lq;
      double fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (VPROBE == 3) {                                  // timing only: operands from registers, no LDS reads
          fa[i] = 1.0 + kq * 1e-9;
          fb[i] = 2.0 + i;
        } else {
          fa[i] = ldsA[cur][kq][wr * 64 + i * 16 + l15];
          fb[i] = ldsB[cur][kq][wc * 64 + i * 16 + l15];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };
  double pc[8], pacc = 0.0;
  if (VPROBE == 1) {
#pragma unroll
    for (int q = 0; q < 8; ++q) pc[q] = B[(long)q * g.ldb + bcol] * 1e-3;       // "candidate coordinates" of this thread's column
  }
  auto valu_probe = [&](int kt) {
    if (VPROBE == 1) {
      const double* __restrict__ xs = g.A + offA + (long)(kt + (wave & 3) * 4) * 8;   // wave-uniform address: scalar loads
#pragma unroll 1
      for (int e = 0; e < 4; ++e) {
        double r2 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const double df = xs[e * 8 + q] - pc[q];
          r2 += df * df;
        }
        pacc += exp(-0.5 * r2);
      }
    }
  };
  if (kend > 0) {
    gload(ra0, rb0, 0);
    if (BK < kend) gload(ra1, rb1, BK);
    lstore(ra0, rb0, 0);
  }
  __syncthreads();
  for (int kt = 0; kt < kend; kt += 2 * BK) {
    if (VPROBE != 4 && kt + 2 * BK < kend) gload(ra0, rb0, kt + 2 * BK);   // (VPROBE 4: timing only, no operand fetch in the loop)
    valu_probe(kt);
    // where a wave stores its share of the next operand tile inside the step is free (the target buffer was last read before
    // the previous barrier): the two waves that share a SIMD (w and w + 4) do it at different points, so one wave's LDS stores
    // and the wait behind them run under the other's MFMAs instead of both stalling the matrix pipe before the barrier
    const int pos = wave >= 4 ? PB : PA;
    const bool on0 = kt < my_kend;
    if (pos == 0 && kt + BK < kend) lstore(ra1, rb1, 1);
    if (on0) compute(0, 0, 2);
    if (pos == 1 && kt + BK < kend) lstore(ra1, rb1, 1);
    if (on0) compute(0, 2, 4);
    if (pos == 2 && kt + BK < kend) lstore(ra1, rb1, 1);
    if (VPROBE != 2) __syncthreads();                    // (VPROBE 2: timing only, no loop barriers)
    if (kt + BK >= kend) break;
    if (VPROBE != 4 && kt + 3 * BK < kend) gload(ra1, rb1, kt + 3 * BK);
    valu_probe(kt + BK);
    const bool on1 = kt + BK < my_kend;
    if (pos == 0 && kt + 2 * BK < kend) lstore(ra0, rb0, 0);
    if (on1) compute(1, 0, 2);
    if (pos == 1 && kt + 2 * BK < kend) lstore(ra0, rb0, 0);
    if (on1) compute(1, 2, 4);
    if (pos == 2 && kt + 2 * BK < kend) lstore(ra0, rb0, 0);
    if (VPROBE != 2) __syncthreads();                    // (VPROBE 2: timing only, no loop barriers)
  }
  if (VPROBE && pacc == 1.2345e300) g.sumsq[0] = pacc;          // never true: keeps the probe's arithmetic alive
  // column sums of squares per 128-row half, fixed order: (wr even) + (wr odd)
  __syncthreads();
  double* red = &ldsA[0][0][0];   // [4 (wr)][128 cols]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double sq = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) sq = __builtin_fma(acc[i][j][r], acc[i][j][r], sq);
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (lq == 0) red[wr * 128 + wc * 64 + j * 16 + l15] = sq;
  }
  __syncthreads();
  if (tid < 256) {
    const int half = tid >> 7, col = tid & 127;
    double* out = g.sumsq + (long)batch * g.strideSumsq + (long)(rtA + half) * g.Ncols + (long)ct * BN;
    out[col] = red[(2 * half) * 128 + col] + red[(2 * half + 1) * 128 + col];
  }
}

// ---------------------------------------------------------------------------------------------
#endif

// Three-buffer form of the 256 x 128 contraction for a TRIANGULAR A (option "swizzle" = 258; the predictive variance:
// A = R upper triangular, row tile rt ends at k = 128 (rt + 1)).  Three findings shape it (tools/mfma_peak_probe.hip,
// profiles/r02/gemm_probes.txt):
//  * integer vector-ALU instructions do NOT run in the shadow of the fp64 matrix pipe on this chip -- every v_add_u32 between
//    MFMAs costs ~3 cycles of MFMA time, a 64-bit multiply-add ~12 -- so the address arithmetic of a GEMM loop is paid in full
//    (6.7 % of the two-buffer kernel).  Here the loop has NO address arithmetic on the vector unit: operand tiles come through
//    buffer loads (wave-uniform row pointer in the resource, advanced on the scalar unit; ONE constant per-thread byte
//    offset), and every LDS access is a constant per-thread base register plus an immediate offset (the loop is unrolled
//    over the 3 buffers x 2 register sets, and the fragments of one lane are laid out contiguously so that they are two
//    ds_read_b128 instead of pairs that the compiler re-bases with a v_add each);
//  * the barrier of a step costs ~5 % when the LDS store drain and the first fragment reads sit next to it: with THREE
//    buffers (159,744 of the CU's 163,840 B) the tile of step t + 1 is already complete in LDS at the barrier that ends
//    step t, so every wave reads its first fragments of tile t + 1 BEFORE that barrier and starts the matrix pipe the moment
//    it is released, and stores its share of tile t + 2 after the first MFMAs of the step are issued;
//  * a wave can issue an fp64 MFMA only every second slot of the matrix pipe (one wave per SIMD reaches half the rate), so
//    (a) a lump of memory instructions between two of a wave's MFMAs loses its slot -- they are spread ONE per MFMA with
//    sched_group_barrier, every step being one straight-line scheduling region -- and (b) a wave that has run out of work
//    cannot be covered by its SIMD partner: the zero blocks below R's diagonal must be shared out EVENLY.  The 256 rows of
//    the workgroup are therefore dealt to the four wave rows in 16-row blocks (wave row wr holds blocks wr, wr + 4, wr + 8,
//    wr + 12), and in the last 16 k-tiles (the diagonal range) every wave skips the blocks that are already zero: 10 tile
//    times instead of 16 for that range.
// Global loads run four tiles ahead (two in registers, two in LDS).  The column sums of squares are accumulated in the order
// of the 128-row kernel -- the partial sum of a lane is handed from wave row to wave row through LDS, because the four
// 16-row blocks that one wave of that kernel squares in sequence sit in four different waves here -- so the results are
// bit-identical to the other tilings.
template <int V>
struct IC {
  static constexpr int value = V;
};

// LDS row strides of the three-buffer kernel: 256 / 128 doubles, NO padding.  Its fragment reads are ds_read_b128 whose four 16-lane
// groups sit in four different k-rows; measured with tools/lds_conflict_probe.sh (profiles/r03/lds_conflict_probe.txt): that pattern is
// conflict-free when the row stride is = 0 mod 256 B and costs exactly twice the LDS cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
// 0.50) when it is = 128 or = 64 mod 256 B -- which the padded strides of the ds_read2_b64 kernels (LDA2 = 272, LDT = 144 doubles:
// = 128 mod 256 B, right for THEIR 8-byte reads) are.  The 36 % conflict cycles of round 2's counters were these reads.
#define LDA3 256
#define LDT3 128
// STORE = 1: the same contraction with the 256 x 128 tile itself written out (alpha * acc, no read of C): the second product of an
// inverse merge, RT21 = -R22^T-form x T'^T, has exactly the variance's shape (A upper triangular, row tile rt ends at 128 (rt + 1)).
template <int STORE>
__global__ __launch_bounds__(512, 1) void gemm_tn_f64_sumsq256x3_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double ldsA[3 * BK * LDA3];    //  98,304 B
  __shared__ __attribute__((aligned(16))) double ldsB[3 * BK * LDT3];     //  49,152 B
  const int nct = g.Ncols / BN;
  const int nrt2 = g.M / BM2;
  const int b = blockIdx.x;
  int rt2, ct, batch;
  if (STORE) {                                         // one-dimensional grid, row-tile-major across the WHOLE batch: every output's heaviest
    const int per = nct * g.batch;                     // row tiles first (a few hundred workgroups of very unequal length: the order is the packing)
    rt2 = b / per;
    const int rem = b - rt2 * per;
    batch = rem / nct;
    ct = rem - batch * nct;
  } else {
    rt2 = b / nct;
    ct = b - rt2 * nct;
    batch = blockIdx.z;
  }
  if (g.rt_desc) rt2 = nrt2 - 1 - rt2;
  const int rtA = 2 * rt2;
  const int kend = BM2 * (rt2 + 1);                    // (launcher: kb = krt = 128, kct = 0, K >= M)
  const int k0 = kend - BM2;                           // the diagonal range [k0, kend): 16 k-tiles
  long offA = (long)batch * g.strideA, offB = (long)batch * g.strideB, offC = (long)batch * g.strideC;
  if (g.batch1 > 0) {                    // two-level batch: z = z2 * batch1 + z1
    const int z1 = batch % g.batch1, z2 = batch / g.batch1;
    offA = (long)z1 * g.strideA + (long)z2 * g.strideA2;
    offB = (long)z1 * g.strideB + (long)z2 * g.strideB2;
    offC = (long)z1 * g.strideC + (long)z2 * g.strideC2;
  }
  const double* A = g.A + offA + (long)rt2 * BM2;
  const double* B = g.B + offB + (long)ct * BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  // Rows: accumulator block q of wave row wr is the 16-row block wr + 4 q of the workgroup's 256 rows.  LDS position of row
  // (wr + 4 q) * 16 + l within a k-row: (q >> 1) * 128 + wr * 32 + l * 2 + (q & 1), so the four values a lane feeds to its
  // four MFMA row blocks are two aligned 16-B pairs (conflict-free b128 reads); columns (wave column wc, block j, lane l):
  // wc * 64 + (j >> 1) * 32 + l * 2 + (j & 1).  Staging map per k-tile: a thread fetches the PAIR of one k-row that is adjacent
  // in LDS (two 8-B loads: rows m, m + 64 of A, columns n, n + 16 of B; 16 lanes per cache line) and stores it with one
  // ds_write_b128; A 16 x 256 = 4 pairs per thread (k-rows + 4 i), B 16 x 128 = 2 (k-rows + 8 i).
  const int sl = tid & 15;
  const int arow = tid >> 7, swr = (tid >> 4) & 3, sh = (tid >> 6) & 1;
  const int brow = tid >> 6, sp = (tid >> 4) & 1, bw = (tid >> 5) & 1;
  const int stA = arow * LDA3 + sh * 128 + swr * 32 + sl * 2, stB = brow * LDT3 + bw * 64 + sp * 32 + sl * 2;
  const int fgA = lq * LDA3 + wr * 32 + l15 * 2, fgB = lq * LDT3 + wc * 64 + l15 * 2;   // this lane's fragment position
  // buffer 2 of A lies beyond the 64-KiB immediate range of the other two: its own base registers (opaque to the compiler,
  // which would otherwise re-derive them with a v_add per access)
  int stA2h = (stA >> 1) + BK * LDA3, fgA2h = (fgA >> 1) + BK * LDA3;        // in 16-B units: the alignment stays visible
  asm volatile("" : "+v"(stA2h), "+v"(fgA2h));
  const int stA2 = stA2h * 2, fgA2 = fgA2h * 2;

  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  v2d ra0[4], rb0[2], ra1[4], rb1[2];
  const unsigned aoff = (unsigned)(((long)arow * g.lda + sh * 128 + swr * 16 + sl) * 8);      // second element: + 64 rows = 512 B
  const unsigned boff = (unsigned)(((long)brow * g.ldb + bw * 64 + sp * 32 + sl) * 8);        // second element: + 16 columns = 128 B
  const int lda8 = g.lda * 8, ldb8 = g.ldb * 8;        // row strides in bytes (rows + 12 at most inside a tile: 32-bit)
  auto fetch = [&](v2d (&ra)[4], v2d (&rb)[2], const double* pa, const double* pb) {
    const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pa), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pb), 0, -1, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i][0] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resA, aoff, 4 * i * lda8, 0));
      ra[i][1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resA, aoff + 512u, 4 * i * lda8, 0));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rb[i][0] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resB, boff, 8 * i * ldb8, 0));
      rb[i][1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resB, boff + 128u, 8 * i * ldb8, 0));
    }
  };
  auto frag = [&](double (&a)[4], double (&bb)[4], auto bufc, auto ksc) {
    constexpr int BUF = decltype(bufc)::value, ROW = BUF * BK + decltype(ksc)::value * 4;
    const double* pa = BUF == 2 ? &ldsA[(ROW - 2 * BK) * LDA3 + fgA2] : &ldsA[ROW * LDA3 + fgA];
    const v2d a01 = *reinterpret_cast<const v2d*>(pa), a23 = *reinterpret_cast<const v2d*>(pa + 128);
    const v2d b01 = *reinterpret_cast<const v2d*>(&ldsB[ROW * LDT3 + fgB]), b23 = *reinterpret_cast<const v2d*>(&ldsB[ROW * LDT3 + fgB + 32]);
    a[0] = a01[0], a[1] = a01[1], a[2] = a23[0], a[3] = a23[1];
    bb[0] = b01[0], bb[1] = b01[1], bb[2] = b23[0], bb[3] = b23[1];
  };
  // Fragments ping-pong between two statically named sets (X: k4-steps 0 and 2, Y: 1 and 3); X of the NEXT tile is read
  // before the barrier that ends the step.
  double xa[4], xb[4], ya[4], yb[4];
  // the interleave of one step: per k4-step NM MFMAs, behind them one by one the 4 fragment reads of the next k4-step and
  // (k4-steps 0..2) store / refill / refill of two staged pairs; with fewer than 10 MFMAs several memory instructions per MFMA
  auto interleave = [&](auto nmc) {
    constexpr int NM = decltype(nmc)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int nmem = ks < 3 ? 10 : 4;
      const int per = NM >= nmem ? 1 : (nmem + NM - 1) / NM;
      int done = 0;
#pragma unroll
      for (int n = 0; n < NM; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int e = 0; e < per; ++e) {
          if (done < nmem) {
            if (done < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (done == 4 || done == 7) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            ++done;
          }
        }
      }
    }
  };
  const long tileA = (long)BK * g.lda, tileB = (long)BK * g.ldb;
  const double* pA = A + (long)(4 * BK) * g.lda;          // rows of the tile the next step fetches (kt + 4 BK, clamped to the last)
  const double* pB = B + (long)(4 * BK) * g.ldb;
  // One step of the full range [0, k0) = ONE straight-line scheduling region: tile kt is in buffer CUR, tile kt + BK in the next
  // buffer, the register set (ra, rb) holds tile kt + 2 BK and is refilled with tile kt + 4 BK.
  auto step = [&](v2d (&ra)[4], v2d (&rb)[2], int kt, auto curc) {
    constexpr int CUR = decltype(curc)::value, NX = (CUR + 1) % 3, NN = (CUR + 2) % 3;
    const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pA), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pB), 0, -1, 0x00020000);
    pA += tileA;                                         // (scalar unit; the full range ends 16 tiles before the contraction does)
    pB += tileB;
    auto mma = [&](const double (&a)[4], const double (&bb)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
    };
    auto stA_ = [&](int i) {
      if (NN == 2) *reinterpret_cast<v2d*>(&ldsA[4 * i * LDA3 + stA2]) = ra[i];
      else *reinterpret_cast<v2d*>(&ldsA[(NN * BK + 4 * i) * LDA3 + stA]) = ra[i];
    };
    auto ldA_ = [&](int i) {
      ra[i][0] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resA, aoff, 4 * i * lda8, 0));
      ra[i][1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resA, aoff + 512u, 4 * i * lda8, 0));
    };
    auto stB_ = [&](int i) {
      *reinterpret_cast<v2d*>(&ldsB[(NN * BK + 8 * i) * LDT3 + stB]) = rb[i];
    };
    auto ldB_ = [&](int i) {
      rb[i][0] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resB, boff, 8 * i * ldb8, 0));
      rb[i][1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resB, boff + 128u, 8 * i * ldb8, 0));
    };
    frag(ya, yb, IC<CUR>(), IC<1>());
    mma(xa, xb);
    stA_(0), ldA_(0), stA_(1), ldA_(1);                  // buffer NN was last read in step kt - BK, before the previous barrier
    frag(xa, xb, IC<CUR>(), IC<2>());
    mma(ya, yb);
    stA_(2), ldA_(2), stA_(3), ldA_(3);
    frag(ya, yb, IC<CUR>(), IC<3>());
    mma(xa, xb);
    stB_(0), ldB_(0), stB_(1), ldB_(1);
    frag(xa, xb, IC<NX>(), IC<0>());                     // complete since the barrier that ended step kt - BK
    mma(ya, yb);
    interleave(IC<16>());
    __builtin_amdgcn_sched_barrier(0);                   // (MFMAs have no memory effect: without this the compiler sinks the last twelve below the barrier)
    __syncthreads();
  };
  // One step of the diagonal range: the same, with the buffer index in a register (the range starts at any of the three)
  // and only the row blocks q >= qmin of this wave still non-zero: 16-row block wr + 4 q is zero from k-tile tau = wr + 4 q + 1
  // of the range on.  The four MFMAs of a block sit behind a wave-uniform branch (an if around MFMAs keeps their accumulators
  // in place; whole-step variants per block count merge 200 live registers five ways and spill), the memory instructions
  // between the blocks.  Past the end of the contraction the step fetches the last tile again and stores into a buffer nobody
  // reads any more.
  auto dstep = [&](v2d (&ra)[4], v2d (&rb)[2], int kt, int cur) {
    const int tau = (kt - k0) / BK;
    const int qmin = (tau - wr + 3) >> 2;                // wave-uniform (wr comes from a readfirstlane)
    const int nx = cur == 2 ? 0 : cur + 1, nn = nx == 2 ? 0 : nx + 1;
    const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pA), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pB), 0, -1, 0x00020000);
    const bool adv = kt + 5 * BK < kend;
    pA += adv ? tileA : 0;
    pB += adv ? tileB : 0;
    const double* fa_c = &ldsA[cur * (BK * LDA3) + fgA];
    const double* fb_c = &ldsB[cur * (BK * LDT3) + fgB];
    const double* fa_n = &ldsA[nx * (BK * LDA3) + fgA];
    const double* fb_n = &ldsB[nx * (BK * LDT3) + fgB];
    double* sa = &ldsA[nn * (BK * LDA3) + stA];
    double* sb = &ldsB[nn * (BK * LDT3) + stB];
    auto fragp = [&](double (&a)[4], double (&bb)[4], const double* pa, const double* pb, int ks) {
      const v2d a01 = *reinterpret_cast<const v2d*>(pa + ks * 4 * LDA3), a23 = *reinterpret_cast<const v2d*>(pa + ks * 4 * LDA3 + 128);
      const v2d b01 = *reinterpret_cast<const v2d*>(pb + ks * 4 * LDT3), b23 = *reinterpret_cast<const v2d*>(pb + ks * 4 * LDT3 + 32);
      a[0] = a01[0], a[1] = a01[1], a[2] = a23[0], a[3] = a23[1];
      bb[0] = b01[0], bb[1] = b01[1], bb[2] = b23[0], bb[3] = b23[1];
    };
    auto stld = [&](int i) {                             // staged pair i: A pairs 0..3, B pairs 4, 5
      if (i < 4) {
        *reinterpret_cast<v2d*>(sa + 4 * i * LDA3) = ra[i];
        ra[i][0] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resA, aoff, 4 * i * lda8, 0));
        ra[i][1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resA, aoff + 512u, 4 * i * lda8, 0));
      } else {
        *reinterpret_cast<v2d*>(sb + 8 * (i - 4) * LDT3) = rb[i - 4];
        rb[i - 4][0] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resB, boff, 8 * (i - 4) * ldb8, 0));
        rb[i - 4][1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(resB, boff + 128u, 8 * (i - 4) * ldb8, 0));
      }
    };
    auto k4 = [&](const double (&a)[4], const double (&bb)[4], int ks) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= qmin) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[q][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bb[j], acc[q][j], 0, 0, 0);
        }
        if (ks < 3 && q < 2) stld(2 * ks + q);
      }
    };
    fragp(ya, yb, fa_c, fb_c, 1);
    k4(xa, xb, 0);
    fragp(xa, xb, fa_c, fb_c, 2);
    k4(ya, yb, 1);
    fragp(ya, yb, fa_c, fb_c, 3);
    k4(xa, xb, 2);
    fragp(xa, xb, fa_n, fb_n, 0);
    k4(ya, yb, 3);
    __syncthreads();
  };
  fetch(ra0, rb0, A, B);                                  // kend >= 16 k-tiles: the prologue needs no guards
  fetch(ra1, rb1, A + tileA, B + tileB);
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<v2d*>(&ldsA[4 * i * LDA3 + stA]) = ra0[i];
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<v2d*>(&ldsB[8 * i * LDT3 + stB]) = rb0[i];
  fetch(ra0, rb0, A + 2 * tileA, B + 2 * tileB);
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<v2d*>(&ldsA[(BK + 4 * i) * LDA3 + stA]) = ra1[i];
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<v2d*>(&ldsB[(BK + 8 * i) * LDT3 + stB]) = rb1[i];
  fetch(ra1, rb1, A + 3 * tileA, B + 3 * tileB);
  __syncthreads();
  frag(xa, xb, IC<0>(), IC<0>());
  // full range: 16 rt2 k-tiles = whole rounds of 6 (3 buffers x 2 register sets, statically named) + 0, 2 or 4 more; single-exit
  // loops only (the accumulators change registers from MFMA to MFMA: every extra exit edge costs a block of copies)
  const int nfull = k0 / BK, rounds = nfull / 6, rest = nfull - 6 * rounds;
  int kt = 0;
  for (int it = 0; it < rounds; ++it, kt += 6 * BK) {
    step(ra0, rb0, kt, IC<0>());
    step(ra1, rb1, kt + BK, IC<1>());
    step(ra0, rb0, kt + 2 * BK, IC<2>());
    step(ra1, rb1, kt + 3 * BK, IC<0>());
    step(ra0, rb0, kt + 4 * BK, IC<1>());
    step(ra1, rb1, kt + 5 * BK, IC<2>());
  }
  if (rest >= 2) {
    step(ra0, rb0, kt, IC<0>());
    step(ra1, rb1, kt + BK, IC<1>());
  }
  if (rest >= 4) {
    step(ra0, rb0, kt + 2 * BK, IC<2>());
    step(ra1, rb1, kt + 3 * BK, IC<0>());
  }
  {
    int cur = rest == 0 ? 0 : (rest == 2 ? 2 : 1);        // the buffer of tile k0
    for (kt = k0; kt < kend; kt += 2 * BK) {
      dstep(ra0, rb0, kt, cur);
      cur = cur == 2 ? 0 : cur + 1;
      dstep(ra1, rb1, kt + BK, cur);
      cur = cur == 2 ? 0 : cur + 1;
    }
  }
  if (STORE) {
    // the tile in whole 1-KiB rows through LDS, 64 rows at a time: pass p holds the 16-row blocks 4 p .. 4 p + 3 = accumulator block q = p of
    // every wave row (row stride 144 doubles: the four k-groups of a wave's ds_write land on disjoint banks)
    double* stage = ldsA;                                 // 64 x 128 doubles
    double* Cout = g.Cout + offC + (long)rt2 * BM2 * g.ldc + (long)ct * BN;
    const double alpha = g.alpha;
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(16 * wr + 4 * r + lq) * LDT3 + wc * 64 + 16 * j + l15] = acc[pass][j][r];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int rr = wave + 8 * e;
        v2d v = *reinterpret_cast<const v2d*>(&stage[rr * LDT3 + lane * 2]);
        v[0] *= alpha;
        v[1] *= alpha;
        *reinterpret_cast<v2d*>(Cout + (long)(64 * pass + rr) * g.ldc + lane * 2) = v;
      }
      __syncthreads();
    }
    return;
  }
  // Column sums of squares per 128-row half in the order of the 128-row kernel: per lane the squares of the 16-row blocks
  // 0..3 (rows + 4 r + lq) of a 64-row group in sequence, then the lanes lq ^ 1, lq ^ 2, then (first 64 rows) + (second).
  // Group q's block i sits in wave row i here: the lane's running sum goes through LDS from wave row to wave row.
  double* xfer = ldsA;                    // [4 (q)][4 (j)][2 (wc)][64 lanes]
  double* red = ldsA + 4 * 4 * 2 * 64;    // [4 (q)][128 cols]
#pragma unroll 1
  for (int stage = 0; stage < 4; ++stage) {
    if (wr == stage) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double* slot = xfer + ((q * 4 + j) * 2 + wc) * 64 + lane;
          double sq = stage == 0 ? 0.0 : *slot;
#pragma unroll
          for (int r = 0; r < 4; ++r) sq = __builtin_fma(acc[q][j][r], acc[q][j][r], sq);   // (explicit: the compiler hoists the squares out of the stage loop otherwise)
          if (stage < 3) {
            *slot = sq;
          } else {
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (lq == 0) red[q * 128 + wc * 64 + j * 16 + l15] = sq;
          }
        }
    }
    __syncthreads();
  }
  if (tid < 256) {
    const int half = tid >> 7, col = tid & 127;
    double* out = g.sumsq + (long)batch * g.strideSumsq + (long)(rtA + half) * g.Ncols + (long)ct * BN;
    out[col] = red[(2 * half) * 128 + col] + red[(2 * half + 1) * 128 + col];
  }
}

void launch_gemm_f64(const GemmArgs& g0, int batch, int epilogue, hipStream_t s) {
  GemmArgs g = g0;
  g.batch = batch;
  const int nrt = g.M / BM, nct = g.Ncols / BN;
  if (nrt == 0 || nct == 0 || batch == 0) return;
  // 256-row tiles (option "swizzle" = 258, the default of large batches): only the plain sum-of-squares contraction from k = 0 with A upper
  // triangular and row tile rt ending at k = 128 (rt + 1) -- the predictive variance; anything else falls back to the 128-row kernel
  if (g.swizzle >= 256) {
    if (epilogue == 1 && !g.prefetch1 && g.M % BM2 == 0 && g.kbeg_rt == 0 && g.kbeg_ct == 0 && g.batch1 == 0) {
      const dim3 grid256((unsigned)((g.M / BM2) * nct), 1, (unsigned)batch);
#ifdef BOCF_PROBES      // the two-buffer kernel of round 2a (256: every wave stores its operand share at the end of a step, 257: waves 4..7 at the start) and its timing-only variants (wrong results)
#define L256(V, A, B) BOCF_LAUNCH((gemm_tn_f64_sumsq256_kernel<V, A, B>), grid256, dim3(512), 0, s, g)
      if (g.swizzle != 258 || g.vprobe) {
        g.stagger = g.swizzle == 257 ? 1 : 0;
        if (g.vprobe == 1) L256(1, 2, 2);
        else if (g.vprobe == 2) L256(2, 2, 2);
        else if (g.vprobe == 3) L256(3, 2, 2);
        else if (g.vprobe == 4) L256(4, 2, 2);
        else if (g.stagger == 1) L256(0, 2, 0);
        else L256(0, 2, 2);
        return;
      }
#undef L256
#endif
      if (g.kb == BM && g.krt == BM && g.kct == 0 && g.K >= g.M) {
        BOCF_LAUNCH((gemm_tn_f64_sumsq256x3_kernel<0>), grid256, dim3(512), 0, s, g);
        return;
      }
    }
    g.swizzle = 0;
  }
  // triangular store products of the variance's shape (A upper triangular with row tile rt ending at k = 128 (rt + 1), no C read):
  // the three-buffer kernel with the store epilogue
  if (epilogue == 0 && !g.no_x3 && !g.Cin && g.M % BM2 == 0 && g.M >= 2 * BM2 && g.kb == BM && g.krt == BM && g.kct == 0 && g.kbeg_rt == 0 && g.kbeg_ct == 0 &&
      g.K >= g.M && !g.upper_only && g.rt_desc && !g.ct_desc) {
    const dim3 grid256((unsigned)((g.M / BM2) * nct * batch), 1, 1);
    BOCF_LAUNCH((gemm_tn_f64_sumsq256x3_kernel<1>), grid256, dim3(512), 0, s, g);
    return;
  }
  if (epilogue == 3) {                                     // store epilogue, square upper-only product: one workgroup per tile on / above the diagonal
    BOCF_LAUNCH((gemm_tn_f64_kernel<3, 1, 8>), dim3((unsigned)(nct * (nct + 1) / 2 * batch)), dim3(512), 0, s, g);
    return;
  }
  dim3 grid((unsigned)(nrt * nct), 1, (unsigned)batch);
#ifdef BOCF_PROBES
  if (g.swizzle >= 100 && (nct % 8 != 0 || nrt % (g.swizzle - 100) != 0)) g.swizzle = 0;
  if (g.swizzle >= 100) {
    grid = dim3((unsigned)(nrt * nct * batch), 1, 1);
  } else if (g.swizzle == 1) {
    const long ns = (long)((nrt + 7) / 8) * ((nct + 7) / 8) * batch;
    grid = dim3((unsigned)(((ns + 7) / 8) * 8 * 64), 1, 1);
  }
#else
  if (g.swizzle != 2) g.swizzle = 0;
#endif
  if (g.swizzle == 2) grid = dim3((unsigned)(nrt * nct * batch), 1, 1);
  // (a two-tile-deep prefetch variant of the store-C kernel, <0, 2>, was measured 2.5x SLOWER: next to the row-staged epilogue
  //  the register allocator spills the prefetch sets inside the k-loop, ~265 VGPRs; a 4-wave workgroup for the store epilogue runs at
  //  half rate whenever it is alone on its CU: removed in round 4)
  if (epilogue == 0)
    BOCF_LAUNCH((gemm_tn_f64_kernel<0, 1, 8>), grid, dim3(512), 0, s, g);
  else if (g.prefetch1)
    BOCF_LAUNCH((gemm_tn_f64_kernel<1, 1>), grid, dim3(256), 0, s, g);
  else
    BOCF_LAUNCH((gemm_tn_f64_kernel<1, 2>), grid, dim3(256), 0, s, g);
}
