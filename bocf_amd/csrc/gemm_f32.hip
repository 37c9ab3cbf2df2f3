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
      const int kq = ks * 4 + lq;
      float fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = lds[cur][0][kq][wr * 64 + i * 16 + l15];
        fb[i] = lds[cur][1][kq][wc * 64 + i * 16 + l15];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // every lane holds 4 rows of ONE column (lane & 15) per accumulator whatever the row map is, so the
  // column sums of squares reduce exactly like the f64 kernel: over regs, over i, then across lane>>4
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
    if (lq == 0) red[wr * 128 + wc * 64 + j * 16 + l15] = s;
  }
  __syncthreads();
  if (tid < 128) {
    double* out = g.sumsq + (long)batch * g.strideSumsq + (long)rt * g.Ncols + (long)ct * BN;
    out[tid] = (double)red[tid] + (double)red[128 + tid];
  }
}

void launch_gemm_f32_sumsq(const GemmArgs32& g, int batch, hipStream_t s) {
  const int nrt = g.M / BM, nct = g.Ncols / BN;
  if (nrt == 0 || nct == 0 || batch == 0) return;
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
