// fp32 MFMA variant of the variance contraction (BASELINE configs[4] asks for fp32):
//   partial[rt][c] = sum_{r in row tile rt} ( sum_{kk < 128(rt+1)} R32[kk][r] * K32[kk][c] )^2
// Same geometry as gemm_f64.hip (128x128 tile, 4 waves, 4x4 accumulators per wave) with
// v_mfma_f32_16x16x4_f32 (32-cycle issue, exact f32 fma chain) and BK = 32, so a k-step is again
// 64 MFMAs x 32 cycles... x2 = 4096 cycles per wave and the LDS footprint stays 72 KiB (2 workgroups/CU).
// Only the sum-of-squares epilogue exists: the fit and the posterior mean stay in fp64.
#include "bocf_internal.h"

typedef float v4f __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#define BK32 32
#define LDT32 144   // padded LDS row (floats): lanes 16..31 of a 32-lane group land 16 banks away

__global__ __launch_bounds__(256, 2) void gemm_tn_f32_sumsq_kernel(GemmArgs32 g) {
  __shared__ float lds[2][2][BK32][LDT32];   // 73,728 B

  const int nct = g.Ncols / BN;
  const int nrt = g.M / BM;
  const int b = blockIdx.x;
  int rt = b / nct;
  const int ct = b - rt * nct;
  rt = nrt - 1 - rt;                          // heavy row tiles first
  const int batch = blockIdx.z;
  int kend = BM * (rt + 1);
  if (kend > g.K) kend = g.K;

  const float* __restrict__ A = g.A + (long)batch * g.strideA + (long)rt * BM;
  const float* __restrict__ B = g.B + (long)batch * g.strideB + (long)ct * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  const int srow = tid >> 5;            // + 8*i : 32 rows of 128 floats per k-tile
  const int scol = (tid & 31) * 4;

  v4f acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  // operand fetch without vector-ALU address arithmetic (gemm_f64.hip): buffer loads, the k-tile's row pointer in the resource
  // (scalar unit), one constant per-thread byte offset, the staged rows as scalar offsets
  v4f ra[4], rb[4];
  const unsigned aoff = (unsigned)(((long)srow * g.lda + scol) * 4), boff = (unsigned)(((long)srow * g.ldb + scol) * 4);
  const int lda32 = (int)(g.lda * 32), ldb32 = (int)(g.ldb * 32);        // 8 rows of floats in bytes
  auto gload = [&](int kt) {
    const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (long)kt * g.lda), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B + (long)kt * g.ldb), 0, -1, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(resA, aoff, i * lda32, 0));
      rb[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(resB, boff, i * ldb32, 0));
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<v4f*>(&lds[buf][0][srow + 8 * i][scol]) = ra[i];
      *reinterpret_cast<v4f*>(&lds[buf][1][srow + 8 * i][scol]) = rb[i];
    }
  };

  if (kend > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < kend; kt += BK32) {
    const bool more = (kt + BK32) < kend;
    if (more) gload(kt + BK32);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      // MFMA block i of a wave row takes rows 4 l + i (lane l), block j of a wave column the columns 4 l + j: the four values a lane
      // feeds to its four blocks are 16 contiguous bytes -- one ds_read_b128 per operand (and the same row / column map as the
      // 256-row kernel below, so that both produce the same sums bit for bit)
      const int kq = ks * 4 + lq;
      const v4f fa = *reinterpret_cast<const v4f*>(&lds[cur][0][kq][wr * 64 + l15 * 4]);
      const v4f fb = *reinterpret_cast<const v4f*>(&lds[cur][1][kq][wc * 64 + l15 * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // every lane holds 4 rows of ONE column (4 (lane & 15) + j) per accumulator, so the column sums of squares reduce like the f64
  // kernel: over regs, over i, then across lane>>4
  __syncthreads();
  float* red = &lds[0][0][0][0];   // [2 (wr)][128 cols]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lq == 0) red[wr * 128 + wc * 64 + 4 * l15 + j] = s;
  }
  __syncthreads();
  if (tid < 128) {
    double* out = g.sumsq + (long)batch * g.strideSumsq + (long)rt * g.Ncols + (long)ct * BN;
    out[tid] = (double)red[tid] + (double)red[128 + tid];
  }
}

// ---------------------------------------------------------------------------------------------
// 256 x 128 tiles, three LDS buffers, eight waves: the fp32 sibling of gemm_tn_f64_sumsq256x3_kernel (gemm_f64.hip; the issue-model
// findings that shape it are described there): no vector-ALU address arithmetic in the loop (buffer loads with scalar row pointers,
// LDS accesses = constant base register + immediate), the next tile's first fragments read before the barrier that ends a step,
// one memory instruction per MFMA slot.  What differs for 4-byte elements:
//   * BK = 32, so that a step is again 8192 MFMA cycles per SIMD and one barrier;
//   * MFMA block i of a wave row takes rows 4 l + i (lane l), block j of a wave column takes columns 4 l + j: the four values a
//     lane feeds to its four blocks are 16 contiguous bytes of the operand row -- ONE ds_read_b128 per operand and k4-step out of
//     a plain row-major LDS image, staged by 16-B loads and 16-B stores with no shuffle.  (Every block then holds rows from all of
//     the wave row's 64, so the zero blocks of R's diagonal range cannot be skipped: the tile runs to the lower half's length.)
// Column sums of squares: per lane over its 16 accumulator rows per column, lanes lq ^ 1, lq ^ 2, the wave rows pairwise -- the 128-row
// kernel above uses the same row / column map and the same order, so the two tilings give the same fp32 sums bit for bit.
#define BM2F 256
template <int V>
struct ICF {
  static constexpr int value = V;
};
__global__ __launch_bounds__(512, 1) void gemm_tn_f32_sumsq256x3_kernel(GemmArgs32 g) {
  __shared__ __attribute__((aligned(16))) float ldsA[3 * BK32 * BM2F];     // 98,304 B
  __shared__ __attribute__((aligned(16))) float ldsB[3 * BK32 * BN];       // 49,152 B
  const int nct = g.Ncols / BN;
  const int nrt2 = g.M / BM2F;
  const int b = blockIdx.x;
  int rt2 = b / nct;
  const int ct = b - rt2 * nct;
  rt2 = nrt2 - 1 - rt2;                                 // heavy row tiles first
  const int batch = blockIdx.z;
  const int kend = BM2F * (rt2 + 1);                    // (launcher: K >= M)
  const float* A = g.A + (long)batch * g.strideA + (long)rt2 * BM2F;
  const float* B = g.B + (long)batch * g.strideB + (long)ct * BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  // staging map per k-tile: A 32 x 256 floats = 4 x 16 B per thread (k-rows + 8 i), B 32 x 128 = 2 x 16 B per thread (k-rows + 16 i)
  const int arow = tid >> 6, acol = (tid & 63) * 4;
  const int brow = tid >> 5, bcol = (tid & 31) * 4;
  const int stA = arow * BM2F + acol, stB = brow * BN + bcol;
  const int fgA = lq * BM2F + wr * 64 + l15 * 4, fgB = lq * BN + wc * 64 + l15 * 4;
  // buffer 2 of A lies beyond the 64-KiB immediate range of the other two: its own base registers, opaque to the compiler
  int stA2q = (stA >> 2) + 2 * BK32 * BM2F / 4, fgA2q = (fgA >> 2) + 2 * BK32 * BM2F / 4;        // in 16-B units: the alignment stays visible
  asm volatile("" : "+v"(stA2q), "+v"(fgA2q));
  const int stA2 = stA2q * 4, fgA2 = fgA2q * 4;

  v4f acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  v4f ra0[4], rb0[2], ra1[4], rb1[2];
  const unsigned aoff = (unsigned)(((long)arow * g.lda + acol) * 4), boff = (unsigned)(((long)brow * g.ldb + bcol) * 4);
  const int lda32 = (int)(g.lda * 32), ldb64 = (int)(g.ldb * 64);          // 8 / 16 rows of floats in bytes
  auto fetch = [&](v4f (&ra)[4], v4f (&rb)[2], const float* pa, const float* pb) {
    const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pa), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pb), 0, -1, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(resA, aoff, i * lda32, 0));
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(resB, boff, i * ldb64, 0));
  };
  auto frag = [&](v4f& a, v4f& bb, auto bufc, auto ksc) {
    constexpr int BUF = decltype(bufc)::value, ROW = BUF * BK32 + decltype(ksc)::value * 4;
    a = BUF == 2 ? *reinterpret_cast<const v4f*>(&ldsA[(ROW - 2 * BK32) * BM2F + fgA2]) : *reinterpret_cast<const v4f*>(&ldsA[ROW * BM2F + fgA]);
    bb = *reinterpret_cast<const v4f*>(&ldsB[ROW * BN + fgB]);
  };
  v4f xa, xb, ya, yb;                                   // fragments ping-pong between two statically named sets
  auto mma = [&](const v4f& a, const v4f& bb) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bb[j], acc[i][j], 0, 0, 0);
  };
  const long tileA = (long)BK32 * g.lda, tileB = (long)BK32 * g.ldb;
  const float* pA = A + 4 * tileA;                       // rows of the tile the next step fetches (kt + 4 BK, clamped to the last)
  const float* pB = B + 4 * tileB;
  // one step = one straight-line scheduling region: tile kt in buffer CUR, tile kt + BK in the next, (ra, rb) holds tile kt + 2 BK and is
  // refilled with tile kt + 4 BK (past the end: the last tile again, stored into a buffer nobody reads any more)
  auto step = [&](v4f (&ra)[4], v4f (&rb)[2], int kt, auto curc) {
    constexpr int CUR = decltype(curc)::value, NX = (CUR + 1) % 3, NN = (CUR + 2) % 3;
    const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pA), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pB), 0, -1, 0x00020000);
    const bool adv = kt + 5 * BK32 < kend;               // (scalar unit)
    pA += adv ? tileA : 0;
    pB += adv ? tileB : 0;
    auto stld = [&](int e) {                             // staged 16-B piece e: A pieces 0..3, B pieces 4, 5
      if (e < 4) {
        if (NN == 2) *reinterpret_cast<v4f*>(&ldsA[8 * e * BM2F + stA2]) = ra[e];
        else *reinterpret_cast<v4f*>(&ldsA[(NN * BK32 + 8 * e) * BM2F + stA]) = ra[e];
        ra[e] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(resA, aoff, e * lda32, 0));
      } else {
        *reinterpret_cast<v4f*>(&ldsB[(NN * BK32 + 16 * (e - 4)) * BN + stB]) = rb[e - 4];
        rb[e - 4] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(resB, boff, (e - 4) * ldb64, 0));
      }
    };
    frag(ya, yb, ICF<CUR>(), ICF<1>()); mma(xa, xb); stld(0);
    frag(xa, xb, ICF<CUR>(), ICF<2>()); mma(ya, yb); stld(1);
    frag(ya, yb, ICF<CUR>(), ICF<3>()); mma(xa, xb); stld(2);
    frag(xa, xb, ICF<CUR>(), ICF<4>()); mma(ya, yb); stld(3);
    frag(ya, yb, ICF<CUR>(), ICF<5>()); mma(xa, xb); stld(4);
    frag(xa, xb, ICF<CUR>(), ICF<6>()); mma(ya, yb); stld(5);
    frag(ya, yb, ICF<CUR>(), ICF<7>()); mma(xa, xb);
    frag(xa, xb, ICF<NX>(), ICF<0>()); mma(ya, yb);       // complete since the barrier that ended step kt - BK
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                     // per k4-step: two fragment reads, (k4-steps 0..5) one store and one refill, one per MFMA
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (ks < 6) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 14, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  };
  fetch(ra0, rb0, A, B);                                  // kend >= 8 k-tiles: the prologue needs no guards
  fetch(ra1, rb1, A + tileA, B + tileB);
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<v4f*>(&ldsA[8 * i * BM2F + stA]) = ra0[i];
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<v4f*>(&ldsB[16 * i * BN + stB]) = rb0[i];
  fetch(ra0, rb0, A + 2 * tileA, B + 2 * tileB);
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<v4f*>(&ldsA[(BK32 + 8 * i) * BM2F + stA]) = ra1[i];
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<v4f*>(&ldsB[(BK32 + 16 * i) * BN + stB]) = rb1[i];
  fetch(ra1, rb1, A + 3 * tileA, B + 3 * tileB);
  __syncthreads();
  frag(xa, xb, ICF<0>(), ICF<0>());
  // 8 (rt2 + 1) k-tiles = whole rounds of 6 (3 buffers x 2 register sets, statically named) + 0, 2 or 4 more; single-exit loops only
  const int nt = kend / BK32, rounds = nt / 6, rest = nt - 6 * rounds;
  int kt = 0;
  for (int it = 0; it < rounds; ++it, kt += 6 * BK32) {
    step(ra0, rb0, kt, ICF<0>());
    step(ra1, rb1, kt + BK32, ICF<1>());
    step(ra0, rb0, kt + 2 * BK32, ICF<2>());
    step(ra1, rb1, kt + 3 * BK32, ICF<0>());
    step(ra0, rb0, kt + 4 * BK32, ICF<1>());
    step(ra1, rb1, kt + 5 * BK32, ICF<2>());
  }
  if (rest >= 2) {
    step(ra0, rb0, kt, ICF<0>());
    step(ra1, rb1, kt + BK32, ICF<1>());
  }
  if (rest >= 4) {
    step(ra0, rb0, kt + 2 * BK32, ICF<2>());
    step(ra1, rb1, kt + 3 * BK32, ICF<0>());
  }
  // column sums of squares: column wc * 64 + 4 c15 + j; rows: the upper 128-row tile = wave rows 0, 1, the lower = wave rows 2, 3
  float* red = ldsA;   // [4 (wr)][128 cols]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) sq += acc[i][j][r] * acc[i][j][r];
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (lq == 0) red[wr * 128 + wc * 64 + 4 * l15 + j] = sq;
  }
  __syncthreads();
  if (tid < 256) {
    const int half = tid >> 7, col = tid & 127;
    double* out = g.sumsq + (long)batch * g.strideSumsq + (long)(2 * rt2 + half) * g.Ncols + (long)ct * BN;
    out[col] = (double)red[(2 * half) * 128 + col] + (double)red[(2 * half + 1) * 128 + col];
  }
}

void launch_gemm_f32_sumsq(const GemmArgs32& g, int batch, hipStream_t s) {
  const int nrt = g.M / BM, nct = g.Ncols / BN;
  if (nrt == 0 || nct == 0 || batch == 0) return;
  if (!g.tile128 && g.M % BM2F == 0 && g.Ncols >= 2048 && g.K >= g.M) {       // (fewer, larger workgroups leave CUs idle on small batches)
    BOCF_LAUNCH(gemm_tn_f32_sumsq256x3_kernel, dim3((unsigned)((g.M / BM2F) * nct), 1, (unsigned)batch), dim3(512), 0, s, g);
    return;
  }
  BOCF_LAUNCH(gemm_tn_f32_sumsq_kernel, dim3((unsigned)(nrt * nct), 1, (unsigned)batch), dim3(256), 0, s, g);
}

__global__ void f64_to_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, long n) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < n) {
    const double2 v = *reinterpret_cast<const double2*>(src + i);
    *reinterpret_cast<float2*>(dst + i) = make_float2((float)v.x, (float)v.y);
  } else if (i < n) {
    dst[i] = (float)src[i];
  }
}

void launch_f64_to_f32(const double* src, float* dst, long n, hipStream_t s) {
  if (n <= 0) return;
  const long threads = (n + 1) / 2;
  BOCF_LAUNCH(f64_to_f32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, src, dst, n);
}
