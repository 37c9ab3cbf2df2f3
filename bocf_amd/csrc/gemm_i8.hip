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
// accumulators, two LDS buffers of one 64-deep chunk each (6 x (8 + 4) KiB) filled through registers one chunk ahead of the reads and two
// ahead of the loads, ONE barrier per chunk; row tile rt stops at k = 128 (rt + 1) (R is triangular), heaviest row tiles first, and workgroups
// that run together have the same row tile(s) -- the same length: they stay in step and share their operands in the L2s (a grid that mixes
// all row tiles runs at the speed of the fp64 kernel).  Epilogue:
// fp64 recombination (smallest group first), squares, and the per-128-row partial sums of the fp64 kernels' layout (fixed order: wave row
// by wave row) -- the finalisation kernels do not know which contraction ran.
#include <hip/hip_runtime.h>

#include "bocf_internal.h"

typedef int v4i_t __attribute__((ext_vector_type(4)));

// weight of digit group g = i + j in the recombination: (1 / 127)^2 254^-g
__device__ __forceinline__ constexpr double i8_weight(int g) {
  double w = 1.0 / (127.0 * 127.0);
  for (int k = 0; k < g; ++k) w /= 254.0;
  return w;
}

// exponents of the columns of the upper triangle of R (m matrices): e[r] = ilogb(max_k |R[k][r]|) + 1  (0 for an all-zero column)
__global__ __launch_bounds__(256) void col_exponent_kernel(const double* __restrict__ R, long strideR, int Np, int* __restrict__ expo) {
  const int j = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  if (r >= Np) return;
  const double* M = R + (long)j * strideR;
  double mx = 0.0;
  for (int k = 0; k <= r; ++k) mx = __builtin_fmax(mx, __builtin_fabs(M[(long)k * Np + r]));
  expo[(long)j * Np + r] = mx > 0.0 ? ilogb(mx) + 1 : 0;
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
  constexpr int RB = 2, CB = 2, WR = 4, WC = 2, NW = 8, TR = 8, TC = 4, NF = NSL * (TR + TC), PER = (NF + NW - 1) / NW;
  __shared__ v4i_t L[2][NSL][TR + TC][64];                  // [buffer][slice][A blocks | B blocks][lane]: 2 x 72 KiB
  __shared__ double red[WR][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv % WR, wc = wv / WR;
  const int nrt = Np / 128, nct = ncols / 64;
  // Workgroups that run at the same time must be the same LENGTH (R is triangular: row tile rt contracts over 128 (rt + 1) rows), or they drift
  // apart and stop sharing their operands in the L2s: the grid is walked in groups of `grp` neighbouring row tiles x all column tiles,
  // heaviest group first (one-dimensional grid: id = (group * nct + ct) * grp + row tile in the group)
  const int grp = group;
  const int gsz = grp * nct, gi = (int)blockIdx.x / gsz, rem = (int)blockIdx.x - gi * gsz;
  const int ct = rem / grp, rt = nrt - 1 - (gi * grp + (rem - ct * grp)), j = blockIdx.z;
  if (rt < 0) return;                                       // (the last group may be short)
  const int rbw = rt * TR, cbw = ct * TC;
  int nch = 2 * (rt + 1);                                   // R is triangular: rows of tile rt see k < 128 (rt + 1)
  if (nch > nchunk) nch = nchunk;
  const size_t sliceA = (size_t)nchunk * rblocks * 1024, sliceB = (size_t)nchunk * cblocks * 1024;     // bytes per slice
  const char* baseA = reinterpret_cast<const char*>(Af) + (size_t)j * NSL * sliceA;
  const char* baseB = reinterpret_cast<const char*>(Bf) + (size_t)j * NSL * sliceB;
  const unsigned loff = lane * 16;
  // fragment f of a chunk, dealt round the waves (f = wv + 8 u): slice f / 12, entry e = f % 12: A block e or B block e - 8
  v4i_t stg[PER];
  auto fetch = [&](int c) {
    const int cc = c < nch ? c : nch - 1;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int f = wv + NW * u;
      if (f < NF) {
        const int sl = f / (TR + TC), e = f - sl * (TR + TC), isB = e >= TR ? 1 : 0;
        // (one buffer resource per slice: a slice stays below 4 GiB, all of K*'s digits do not)
        const char* p = isB ? baseB + (size_t)sl * sliceB : baseA + (size_t)sl * sliceA;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, -1, 0x00020000);
        const unsigned fi = isB ? (unsigned)cc * cblocks + cbw + (e - TR) : (unsigned)cc * rblocks + rbw + e;
        stg[u] = __builtin_bit_cast(v4i_t, __builtin_amdgcn_raw_buffer_load_b128(rs, loff, (int)(fi * 1024u), 0));
      }
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int f = wv + NW * u;
      if (f < NF) {
        const int sl = f / (TR + TC), e = f - sl * (TR + TC);
        L[buf][sl][e][lane] = stg[u];
      }
    }
  };
  v4i_t acc[NSL][RB][CB];
#pragma unroll
  for (int g = 0; g < NSL; ++g)
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int jb = 0; jb < CB; ++jb) acc[g][i][jb] = (v4i_t){0, 0, 0, 0};
  fetch(0);
  put(0);
  fetch(1);
  __syncthreads();
  auto chunk = [&](int buf, int c) {
    // buffer buf ^ 1 was read in chunk c - 1 (the barrier at its end): chunk c + 1 goes in, chunk c + 2 into the staging registers
    put(buf ^ 1);
    fetch(c + 2);
    v4i_t a[NSL][RB], b[NSL][CB];
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
#pragma unroll
      for (int jb = 0; jb < CB; ++jb) b[s][jb] = L[buf][s][TR + wc * CB + jb][lane];
#pragma unroll
      for (int i = 0; i < RB; ++i) a[s][i] = L[buf][s][wr * RB + i][lane];
    }
#pragma unroll
    for (int g = 0; g < NSL; ++g)
#pragma unroll
      for (int s = 0; s <= g; ++s)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int jb = 0; jb < CB; ++jb) acc[g][i][jb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s][i], b[g - s][jb], acc[g][i][jb], 0, 0, 0);
    __syncthreads();
  };
#pragma unroll 1
  for (int c = 0; c < nch; c += 2) {                        // (nch is even)
    chunk(0, c);
    chunk(1, c + 1);
  }
  // fp64 recombination; element r of lane l is row 4 (l / 16) + r, column l % 16 of its block.  The squares of this wave's 32 rows are summed
  // per column (4 r, 2 blocks, then the four 16-lane groups), the four wave rows in order through LDS
  const int q = lane >> 4;
  const int colscale = eB[j];
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
}

size_t i8_operand_bytes(int Np, int ncols, int m) { return (size_t)m * BOCF_I8_SLICES * (size_t)(Np / 64) * (size_t)(ncols / 16) * 1024; }

void launch_col_exponents(const double* R, long strideR, int Np, int* expo, int m, hipStream_t s) {
  BOCF_LAUNCH(col_exponent_kernel, dim3((unsigned)((Np + 255) / 256), (unsigned)m), dim3(256), 0, s, R, strideR, Np, expo);
}

void launch_slice_operand(const double* X, long ld, long strideX, int krows, int Np, int ncols, const int* expo, int expo_stride, void* F, int m,
                          hipStream_t s) {
  BOCF_LAUNCH((slice_operand_kernel<BOCF_I8_SLICES>), dim3((unsigned)((ncols / 16 + 3) / 4), (unsigned)(Np / 64), (unsigned)m), dim3(256), 0, s, X, ld,
              strideX, krows, ncols, expo, expo_stride, static_cast<v4i_t*>(F), Np / 64);
}

// sumsq[j][rt][col] = sum over the 128 rows of tile rt of V[r][col]^2, V = R^T K* from the digit fragments (ncols a multiple of 64)
void launch_var_i8(const void* Af, const void* Bf, int Np, int ncols, const int* eA, const int* eB, double* sumsq, long strideSumsq, int m,
                   hipStream_t s, int group) {
  const int nrt = Np / 128, nct = ncols / 64;
  if (group < 1) group = ncols >= 32768 ? 2 : 1;            // measured at N = 4096: 65 536 columns 46.9 / 47.4 / 49.2 / 68 ms for 2 / 1 / 4 / 32; 8192 columns 5.71 / 5.57 / 6.08
  if (group > nrt) group = nrt;
  const int ngroups = (nrt + group - 1) / group;
  BOCF_LAUNCH((var_i8_kernel<BOCF_I8_SLICES>), dim3((unsigned)(ngroups * group * nct), 1, (unsigned)m), dim3(512), 0, s,
              static_cast<const v4i_t*>(Af), static_cast<const v4i_t*>(Bf), Np / 64, Np / 16, ncols / 16, eA, eB, Np, sumsq, strideSumsq, ncols, group);
}
