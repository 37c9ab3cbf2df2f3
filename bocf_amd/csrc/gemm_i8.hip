// The predictive variance's contraction  v = L^-1 K(X, x*),  var = k** - sum v^2  (GPy/inference/latent_function_inference/posterior.py:308-313)
// in EXACT int8 arithmetic (option "predict_i8"; the Ozaki scheme): the fp64 MFMA kernel (gemm_f64.hip) runs at 0.93 of its pipe, the int8
// pipe of this chip is 62 x faster, and a contraction cut into 6 x 6 / 2 int8 products is still the same sum to 4e-12.
//
//   V[r][c] = sum_k R[k][r] K*[k][c]                R upper triangular (k <= r), K* = K(X, x*) of one candidate chunk
//
// Every operand COLUMN (fixed r resp. c, all k) gets one power-of-two scale and is cut into NSL = 6 signed digits of radix 254 -- all an int8
// holds: |d_i| <= 127, 47.9 bits in six digits (radix 128 gave 41: 1.3e-4 instead of 2e-6 relative on config 3's variances of 1e-6):
//   a = 2^e / 127 sum_i d_i 254^-i,  d_0 = rint(127 a 2^-e), the remainder (exact) times 254 gives the next digit.
// R's columns take their own maximum (a pass over R per fit: slice_operand's exponents), K*'s columns all take the kernel's variance
// (a stationary kernel never exceeds it: no reduction).  The 21 digit products with i + j <= 5 are exact in int32 -- 2^14 per term, 2^26 per
// product over K = 4096, 2^28.6 per group g = i + j (up to N = 16384: the launcher's caller keeps larger models on the fp64 kernel) -- and
// V = 2^(e_r + e_c) / 127^2 sum_g 254^-g G_g; what is dropped (i + j >= 6) is below 254^-6 = 4e-15 of the column scales per term.  Measured (tools/ozaki_tile_probe.hip, 4096^3): 0.98 ms against 1.88 ms
// for the fp64 kernel, 2e-11 absolute on values of size 5, 3e-13 relative on the column sums of squares.
//
// Layout: the digits are stored FRAGMENT by fragment of v_mfma_i32_16x16x64_i8 -- [slice][k chunk of 64][block of 16 columns][lane][16 bytes],
// byte t of lane l = k 16 (l / 16) + t of column l % 16 (found by trial in the probe) -- so a wave's operand is ONE 16-byte load per lane, and
// the kernel: a 128 x 64 piece per workgroup, 8 waves (two per SIMD: one alone reaches a third of the int8 rate) of 2 x 2 blocks x 6 group
// accumulators, two LDS buffers of one 64-deep chunk each (6 x (8 + 4) KiB) filled through registers one step ahead of the reads and two
// ahead of the loads, ONE barrier per step, and inside a step the LDS reads of the NEXT step under this step's matrix instructions (digit
// by digit into the registers the last product of a digit has left).  Row tile rt stops at k = 128 (rt + 1) (R is triangular): a workgroup
// takes the light tile p and then the heavy tile nrt - 1 - p of one column tile -- all workgroups the same length, so they stay in step and
// share their operands in the L2s (a grid that mixes lengths runs at the speed of the fp64 kernel) --, and the 32 workgroups an XCD runs
// together are a block of 4 pairs x 8 column tiles.  Epilogue per row tile:
// fp64 recombination (smallest group first), squares, and the per-128-row partial sums of the fp64 kernels' layout (fixed order: wave row
// by wave row) -- the finalisation kernels do not know which contraction ran.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "bocf_internal.h"

typedef int v4i_t __attribute__((ext_vector_type(4)));

// weight of digit group g = i + j in the recombination: (1 / 127)^2 254^-g
__device__ __forceinline__ constexpr double i8_weight(int g) {
  double w = 1.0 / (127.0 * 127.0);
  for (int k = 0; k < g; ++k) w /= 254.0;
  return w;
}

// exponents of the columns of the upper triangle of R (m matrices): e[r] = ilogb(max_k |R[k][r]|) + 1.  One workgroup per 256 columns x 256
// rows on or above the diagonal, the blocks of a column meet in an atomic maximum (expo starts at the memset pattern 0x80808080, far below
// any exponent; an all-zero column keeps it: its digits are zero whatever the scale)
__global__ __launch_bounds__(256) void col_exponent_kernel(const double* __restrict__ R, long strideR, int Np, int* __restrict__ expo) {
  const int j = blockIdx.z, r = blockIdx.x * 256 + threadIdx.x, k0 = blockIdx.y * 256;
  if (blockIdx.y > blockIdx.x || r >= Np) return;
  const double* M = R + (long)j * strideR;
  const int k1 = k0 + 255 < r ? k0 + 255 : r;
  double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
  int k = k0;
  for (; k + 3 <= k1; k += 4) {
    m0 = __builtin_fmax(m0, __builtin_fabs(M[(long)k * Np + r]));
    m1 = __builtin_fmax(m1, __builtin_fabs(M[(long)(k + 1) * Np + r]));
    m2 = __builtin_fmax(m2, __builtin_fabs(M[(long)(k + 2) * Np + r]));
    m3 = __builtin_fmax(m3, __builtin_fabs(M[(long)(k + 3) * Np + r]));
  }
  for (; k <= k1; ++k) m0 = __builtin_fmax(m0, __builtin_fabs(M[(long)k * Np + r]));
  const double mx = __builtin_fmax(__builtin_fmax(m0, m1), __builtin_fmax(m2, m3));
  if (mx > 0.0) atomicMax(expo + (long)j * Np + r, ilogb(mx) + 1);
}

// X[k][col] (k-major, leading dimension ld, m matrices) -> digit fragments.  One wave = one fragment position (chunk kc, column block cb):
// lane l reads its 16 k's of its column and writes 16 bytes per slice.  expo: per column (expo_stride = columns per matrix) or, with
// expo_stride = 0, one exponent per matrix.  Rows >= krows (padding) read as zero.
template <int NSL>
__global__ __launch_bounds__(256) void slice_operand_kernel(const double* __restrict__ X, long ld, long strideX, int krows, int ncols,
                                                            const int* __restrict__ expo, int expo_stride, v4i_t* __restrict__ F, int nchunk) {
  const int j = blockIdx.z, kc = blockIdx.y, cb = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int cblocks = ncols / 16;
  if (cb >= cblocks) return;
  const int q = lane >> 4, col = 16 * cb + (lane & 15);
  const int e = expo_stride ? expo[(long)j * expo_stride + col] : expo[j];
  const double* src = X + (long)j * strideX + col;
  unsigned w[NSL][4];
#pragma unroll
  for (int s = 0; s < NSL; ++s) w[s][0] = w[s][1] = w[s][2] = w[s][3] = 0u;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int k = 64 * kc + 16 * q + t;
    double x = k < krows ? __builtin_ldexp(src[(long)k * ld], -e) * 127.0 : 0.0;  // |x| < 127
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const double dg = __builtin_rint(x);                  // |dg| <= 127; x - dg is exact, |x - dg| <= 1/2
      w[s][t >> 2] |= ((unsigned)(int)dg & 0xffu) << (8 * (t & 3));
      x = (x - dg) * 254.0;
    }
  }
  const size_t per_slice = (size_t)nchunk * cblocks * 64;
  v4i_t* out = F + (size_t)j * NSL * per_slice + ((size_t)kc * cblocks + cb) * 64 + lane;
#pragma unroll
  for (int s = 0; s < NSL; ++s) out[(size_t)s * per_slice] = (v4i_t){(int)w[s][0], (int)w[s][1], (int)w[s][2], (int)w[s][3]};
}

template <int NSL>
__global__ __launch_bounds__(512, 1) void var_i8_kernel(const v4i_t* __restrict__ Af, const v4i_t* __restrict__ Bf, int nchunk, int rblocks, int cblocks,
                                                        const int* __restrict__ eA, const int* __restrict__ eB, int Np, double* __restrict__ sumsq,
                                                        long strideSumsq, int ncols, int group) {
  constexpr int RB = 2, CB = 2, WR = 4, WC = 2, NW = 8, TR = 8, TC = 4, NFA = NSL, NFB = NSL * TC / NW, PER = NFA + NFB;
  constexpr int BUFB = NSL * (TR + TC) * 1024, BOFF = NSL * TR * 1024;      // bytes per buffer (72 KiB); where its B fragments start (48 KiB)
  static_assert(TR == NW && (NSL * TC) % NW == 0 && RB == 2 && CB == 2, "fragments are dealt evenly to the waves");
  // one buffer: [slice][8 A blocks][lane] then [slice][4 B blocks][lane], 16 bytes per lane -- every LDS access of a step is one base
  // register plus an immediate below 64 KiB
  __shared__ v4i_t L[2 * BUFB / 16];
  __shared__ double red[WR][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv % WR, wc = wv / WR;
  const int nrt = Np / 128, nct = ncols / 64, npair = (nrt + 1) / 2;
  // R is triangular: row tile rt contracts over 128 (rt + 1) rows = 2 (rt + 1) chunks.  Workgroups that run at the same time must be the same
  // LENGTH, or they drift apart and stop sharing their operands in the L2s (a grid that mixes lengths runs at the fp64 kernel's speed): a
  // workgroup takes a PAIR of row tiles of one column tile -- the light tile p first, its chunks in ascending order, then the heavy tile
  // nrt - 1 - p with its chunks in DESCENDING order -- so every workgroup is 2 nrt + 2 chunks long and at step i of that sequence every pair
  // reads chunk i (still in its light tile) or chunk 2 nrt + 1 - i (in its heavy one) of K*'s digits, whatever p is: the workgroups of one
  // column tile keep reading the same two chunks.  One pipeline of loads runs through both tiles.  (Odd nrt: the middle tile alone.)
  // The grid is one-dimensional, pair fastest: id = ct * npair + pair, `group` > 1 walks it in bands of `group` pairs x all column tiles.
  int ct, pr;
  const int j = blockIdx.z;
  if (group > 0) {
    const int gsz = group * nct, gi = (int)blockIdx.x / gsz, rem = (int)blockIdx.x - gi * gsz;
    ct = rem / group;
    pr = gi * group + (rem - ct * group);
  } else {
    // workgroup id -> XCD id % 8 (the dispatcher deals them round): the 32 workgroups an XCD runs together are a block of 4 pairs x 8 column
    // tiles -- the fewest distinct operand bytes (4 x 48 + 8 x 24 KiB per chunk) for its L2
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, w = slot & 31, t = (slot >> 5) * 8 + xcd;
    const int PG = (npair + 3) / 4, pgi = t % PG, cg = t / PG;
    pr = 4 * pgi + (w & 3);
    ct = 8 * cg + (w >> 2);
    if (ct >= nct) return;
  }
  if (pr >= npair) return;                                  // (the last band may be short)
  const int rtL = pr, rtH = nrt - 1 - pr;
  const int nchL = 2 * (rtL + 1), ntot = rtH > rtL ? 2 * nrt + 2 : nchL;      // (both even)
  const int cbw = ct * TC;
  const size_t sliceA = (size_t)nchunk * rblocks * 1024, sliceB = (size_t)nchunk * cblocks * 1024;     // bytes per slice
  const char* baseA = reinterpret_cast<const char*>(Af) + (size_t)j * NSL * sliceA;
  const char* baseB = reinterpret_cast<const char*>(Bf) + (size_t)j * NSL * sliceB;
  const unsigned loff = lane * 16;
  // wave w stages A block w of every slice (u = 0 .. 5: LDS byte u * 8 KiB + w KiB) and the B fragments w + 8 v of the buffer's
  // [slice][4] order (v = 0 .. 2: slice 2 v + w / 4, block w % 4, LDS byte 48 KiB + w KiB + v * 8 KiB)
  v4i_t stg[PER];
  const int bsl = wv >> 2, bblk = wv & 3;
  auto fetch = [&](int i) {                                 // step i of the pair's sequence
    const int ii = i < ntot ? i : ntot - 1;
    const int heavy = ii >= nchL ? 1 : 0;
    const int cc = heavy ? ntot - 1 - ii : ii, rbw = (heavy ? rtH : rtL) * TR;
    // (one buffer resource per slice: a slice stays below 4 GiB, all of K*'s digits do not)
    const unsigned fa = ((unsigned)cc * rblocks + rbw + wv) * 1024u, fb = ((unsigned)cc * cblocks + cbw + bblk) * 1024u;
#pragma unroll
    for (int u = 0; u < NFA; ++u) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(baseA + (size_t)u * sliceA), 0, -1, 0x00020000);
      stg[u] = __builtin_bit_cast(v4i_t, __builtin_amdgcn_raw_buffer_load_b128(rs, loff, (int)fa, 0));
    }
#pragma unroll
    for (int v = 0; v < NFB; ++v) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(baseB + (size_t)(2 * v + bsl) * sliceB), 0, -1, 0x00020000);
      stg[NFA + v] = __builtin_bit_cast(v4i_t, __builtin_amdgcn_raw_buffer_load_b128(rs, loff, (int)fb, 0));
    }
  };
  char* lbase = reinterpret_cast<char*>(&L[0]);
  const unsigned offW = loff + wv * 1024;
  auto put = [&](unsigned o, int buf) {                     // o: offW, or a copy of it the compiler cannot see through
#pragma unroll
    for (int u = 0; u < NFA; ++u) *reinterpret_cast<v4i_t*>(lbase + (o + (unsigned)(buf * BUFB + u * 8192))) = stg[u];
#pragma unroll
    for (int v = 0; v < NFB; ++v) *reinterpret_cast<v4i_t*>(lbase + (o + (unsigned)(buf * BUFB + BOFF + v * 8192))) = stg[NFA + v];
  };
  v4i_t acc[NSL][RB][CB];
  auto clear = [&]() {
#pragma unroll
    for (int g = 0; g < NSL; ++g)
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int jb = 0; jb < CB; ++jb) acc[g][i][jb] = (v4i_t){0, 0, 0, 0};
  };
  clear();
  static_assert(NSL == 6, "the product order below is written out for six digits");
  // Software pipeline, one barrier per step: while the 84 matrix instructions of step k run on the operand registers, the wave reads the
  // operands of step k + 1 out of LDS buffer (k + 1) & 1 INTO THE SAME REGISTERS, digit by digit as soon as the last product that needs a
  // digit is through (the products without digit 0 first -- then the writes and the loads --, then (0, d), (d, 0) for d = 5 ... 1, then
  // (0, 0): every digit is reloaded at least 16 matrix instructions before its first use in the next step, except digit 0 at the barrier),
  // writes the staged fragments of step k + 2 into buffer k & 1 (read in step k - 1) and loads step k + 3 from memory.  LDS traffic
  // (24 KiB read and 9 KiB written per wave and step) then runs underneath the matrix pipe instead of in a phase of its own between
  // two barriers.  All 256 registers a wave has at two waves per SIMD: 96 accumulators, 96 operands, 36 staged -- no scratch.
  v4i_t a[NSL][RB], b[NSL][CB];
  // LDS addresses are a byte offset the compiler cannot see through (the empty asm in step()) plus a constant: with the addresses of a
  // step as loop invariants it keeps them in registers it does not have, and reloads them from scratch behind a vmcnt(0)
  const int sofA = wr * RB * 1024, sofB = BOFF + wc * CB * 1024, sofW = wv * 1024;      // (wave-uniform: scalar registers)
  const unsigned offA = loff + sofA, offB = loff + sofB;
  auto rdA = [&](unsigned o, int sl) {
#pragma unroll
    for (int i = 0; i < RB; ++i) a[sl][i] = *reinterpret_cast<const v4i_t*>(lbase + (o + (unsigned)(sl * TR * 1024 + i * 1024)));
  };
  auto rdB = [&](unsigned o, int sl) {
#pragma unroll
    for (int jb = 0; jb < CB; ++jb) b[sl][jb] = *reinterpret_cast<const v4i_t*>(lbase + (o + (unsigned)(sl * TC * 1024 + jb * 1024)));
  };
  auto prod = [&](int sa, int sb) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int jb = 0; jb < CB; ++jb) acc[sa + sb][i][jb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[sa][i], b[sb][jb], acc[sa + sb][i][jb], 0, 0, 0);
  };
  // nothing moves across a pin -- memory accesses (memory clobber) and matrix instructions (it names every accumulator): a pin on either
  // side keeps the reads of a digit where they are written, between its last product and the products that follow
#define I8_ACC(g) "+v"(acc[g][0][0]), "+v"(acc[g][0][1]), "+v"(acc[g][1][0]), "+v"(acc[g][1][1])
#define I8_PIN(g) asm volatile("" : I8_ACC(0), I8_ACC(1), I8_ACC(2), I8_ACC(3), I8_ACC(4), I8_ACC(5) : : "memory")
  auto step = [&](int k) {
    const int P = k & 1;                                    // the buffer step k + 2 goes to
    unsigned lo = loff;                                     // (the one lane-dependent value that lives across the loop)
    asm volatile("" : "+v"(lo));
    const unsigned oa = lo + (unsigned)(sofA + (P ^ 1) * BUFB), ob = lo + (unsigned)(sofB + (P ^ 1) * BUFB), ow = lo + (unsigned)(sofW + P * BUFB);
    prod(2, 3); prod(3, 2); prod(2, 2);
    prod(1, 4); prod(4, 1); prod(1, 3); prod(3, 1); prod(1, 2); prod(2, 1); prod(1, 1);
    I8_PIN(2);
    put(ow, 0);
    fetch(k + 3);
    I8_PIN(2);
    prod(0, 5); prod(5, 0);
    I8_PIN(5);
    rdA(oa, 5); rdB(ob, 5);
    I8_PIN(5);
    prod(0, 4); prod(4, 0);
    I8_PIN(4);
    rdA(oa, 4); rdB(ob, 4);
    I8_PIN(4);
    prod(0, 3); prod(3, 0);
    I8_PIN(3);
    rdA(oa, 3); rdB(ob, 3);
    I8_PIN(3);
    prod(0, 2); prod(2, 0);
    I8_PIN(2);
    rdA(oa, 2); rdB(ob, 2);
    I8_PIN(2);
    prod(0, 1); prod(1, 0);
    I8_PIN(1);
    rdA(oa, 1); rdB(ob, 1);
    I8_PIN(1);
    prod(0, 0);
    I8_PIN(0);
    rdA(oa, 0); rdB(ob, 0);
    // (no fence: the loads from memory stay in flight across the barrier; LDS reads and writes of this wave are complete.  The
    // accumulators are named so that no product of the next step is moved up in front of the barrier, behind this step's reads)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : I8_ACC(0), I8_ACC(1), I8_ACC(2), I8_ACC(3), I8_ACC(4), I8_ACC(5) : : "memory");
  };
#undef I8_PIN
#undef I8_ACC
  fetch(0);
  put(offW, 0);
  fetch(1);
  put(offW, 1);
  fetch(2);
  __syncthreads();
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    rdA(offA, sl);
    rdB(offB, sl);
  }
  __syncthreads();                                          // buffer 0 is free for step 2
  // fp64 recombination of a finished row tile; element r of lane l is row 4 (l / 16) + r, column l % 16 of its block.  The squares of this
  // wave's 32 rows are summed per column (4 r, 2 blocks, then the four 16-lane groups), the four wave rows in order through LDS
  const int q = lane >> 4;
  const int colscale = eB[j];
  auto finish = [&](int rt) {
#pragma unroll
    for (int jb = 0; jb < CB; ++jb) {
      double ss = 0.0;
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double v = 0.0;
#pragma unroll
          for (int g = NSL - 1; g >= 0; --g) v += (double)acc[g][i][jb][r] * i8_weight(g);
          const int row = 128 * rt + 16 * (wr * RB + i) + 4 * q + r;
          v = __builtin_ldexp(v, eA[(long)j * Np + row] + colscale);
          ss += v * v;
        }
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (lane < 16) red[wr][wc * 32 + jb * 16 + lane] = ss;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int col = ct * 64 + threadIdx.x;
      if (col < ncols) sumsq[(long)j * strideSumsq + (long)rt * ncols + col] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
  };
#pragma unroll 1
  for (int c = 0; c < ntot; ++c) {
    step(c);
    if (c + 1 == nchL) {                                    // the light tile is complete: its sums out, on with the heavy one
      finish(rtL);
      clear();
      __syncthreads();                                      // (red is written again at the end)
    }
  }
  if (ntot > nchL) finish(rtH);
}

size_t i8_operand_bytes(int Np, int ncols, int m) { return (size_t)m * BOCF_I8_SLICES * (size_t)(Np / 64) * (size_t)(ncols / 16) * 1024; }

// (expo must hold the byte pattern 0x80 on entry: the caller's memset)
void launch_col_exponents(const double* R, long strideR, int Np, int* expo, int m, hipStream_t s) {
  const unsigned nb = (unsigned)((Np + 255) / 256);
  BOCF_LAUNCH(col_exponent_kernel, dim3(nb, nb, (unsigned)m), dim3(256), 0, s, R, strideR, Np, expo);
}

void launch_slice_operand(const double* X, long ld, long strideX, int krows, int Np, int ncols, const int* expo, int expo_stride, void* F, int m,
                          hipStream_t s) {
  BOCF_LAUNCH((slice_operand_kernel<BOCF_I8_SLICES>), dim3((unsigned)((ncols / 16 + 3) / 4), (unsigned)(Np / 64), (unsigned)m), dim3(256), 0, s, X, ld,
              strideX, krows, ncols, expo, expo_stride, static_cast<v4i_t*>(F), Np / 64);
}

// sumsq[j][rt][col] = sum over the 128 rows of tile rt of V[r][col]^2, V = R^T K* from the digit fragments (ncols a multiple of 64)
void launch_var_i8(const void* Af, const void* Bf, int Np, int ncols, const int* eA, const int* eB, double* sumsq, long strideSumsq, int m,
                   hipStream_t s, int group) {
  const int nrt = Np / 128, nct = ncols / 64, npair = (nrt + 1) / 2;
  unsigned grid;
  if (group < 1) {
    const int tiles = ((npair + 3) / 4) * ((nct + 7) / 8);
    grid = (unsigned)((tiles + 7) / 8) * 256u;
    group = 0;
  } else {
    if (group > npair) group = npair;
    grid = (unsigned)(((npair + group - 1) / group) * group * nct);
  }
  BOCF_LAUNCH((var_i8_kernel<BOCF_I8_SLICES>), dim3(grid, 1, (unsigned)m), dim3(512), 0, s,
              static_cast<const v4i_t*>(Af), static_cast<const v4i_t*>(Bf), Np / 64, Np / 16, ncols / 16, eA, eB, Np, sumsq, strideSumsq, ncols, group);
}
