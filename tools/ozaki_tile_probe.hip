// The int8 (Ozaki) TILE kernel of VERDICT r2 item 8 / r3 item 7: can an error-free int8 split beat the fp64-MFMA variance contraction
// (posterior.py:308-313: v = L^-1 K*, var = k** - sum v^2), which runs at 0.93 of the 78.6 TFLOP/s peak?
//
//   C[m][n] = sum_k A[k][m] B[k][n],  K = 4096          (A = a 256-row slab of R^T, B = a 128-candidate slab of K*)
//
// Every operand column (fixed m resp. n, all k) gets ONE power-of-two scale and is cut into 7 signed 7-bit digits (int8 "slices"),
// a = 2^e sum_i d_i 2^(-6 - 7 i); the 28 slice products with i + j <= 6 are EXACT in int32 (|d| <= 64: 2^12 per term, 2^24 per
// product over K = 4096, at most 2^27 per group g = i + j), and C = 2^(e_m + f_n) sum_g 2^(-12 - 7 g) G_g with G_g = sum_{i+j=g} A_i^T B_j.
// What is measured: the tile kernel alone -- slices already in memory, laid out fragment by fragment so that a lane's 16 bytes of a
// v_mfma_i32_16x16x64_i8 operand are one 16-byte load -- against the time the product's fp64 kernel needs for the same tile
// (2 x 256 x 128 x 4096 flop at 0.93 x 78.6 TFLOP/s / 256 CUs = 0.94 ms per compute unit), the error against a long-double sum, and
// the error of the fused epilogue's sum of squares.  One wave per SIMD (512 registers): a wave keeps 4 x 2 blocks of 16 x 16 outputs
// x 7 group accumulators (224 registers) and, per 64-deep k chunk, the 7 x (4 + 2) operand fragments (168 registers); the operand traffic is
// 42 KiB per 224 matrix instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/ozaki_tile_probe.hip -o tools/ozaki_tile_probe.bin && ./tools/ozaki_tile_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define NS 7              // slices
#define KC 64             // k per matrix instruction

// ---- operand layout of v_mfma_i32_16x16x64_i8, found by trial: one instruction on random bytes
__global__ void one_mfma(const v4i* a, const v4i* b, v4i* d) {
  d[threadIdx.x] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], (v4i){0, 0, 0, 0}, 0, 0, 0);
}
// hypothesis h: byte t of lane l is k = kmap(h, l / 16, t) of row / column l % 16
static int kmap(int h, int q, int t) { return h == 0 ? 16 * q + t : 8 * q + (t & 7) + 32 * (t >> 3); }
// output hypothesis o: element r of lane l is row 4 (l / 16) + r (o = 0) or 4 r + l / 16 (o = 1), column l % 16
static int rmap(int o, int q, int r) { return o == 0 ? 4 * q + r : 4 * r + q; }

// ---- the tile kernel.  Fragments: Af[slice][chunk][row block][lane] (16 bytes each), Bf[slice][chunk][column block][lane]; the kernel never
// needs to know which (row, k) a byte is -- the host packs the fragments with the layout found above, and hands the scales over in the
// accumulator's own order: rs[row block][lane][r], cs[column block][lane].  One workgroup = 4 waves (2 x 2) = a 128 x 64 piece.  Epilogue:
// fp64 recombination (smallest group first), the value out (in fragment order) and its square into the column's sum (the variance's
// sum over the training rows).
template <int RB, int CB, int WR, int WC>                  // a wave: RB x CB blocks of 16 x 16; a workgroup: WR x WC waves
__global__ __launch_bounds__(64 * WR * WC, 1) void ozaki_tile(const v4i* __restrict__ Af, const v4i* __restrict__ Bf, int nchunk, int rblocks, int cblocks,
                                                             const double* __restrict__ rs, const double* __restrict__ cs, double* __restrict__ Cf,
                                                             double* __restrict__ colss) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rb0 = (blockIdx.x * WR + (wv % WR)) * RB, cb0 = (blockIdx.y * WC + (wv / WR)) * CB;
  v4i acc[NS][RB][CB];
#pragma unroll
  for (int g = 0; g < NS; ++g)
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[g][i][j] = (v4i){0, 0, 0, 0};
#pragma unroll 1
  for (int c = 0; c < nchunk; ++c) {
    v4i a[NS][RB], b[NS][CB];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < RB; ++i) a[s][i] = Af[(((size_t)s * nchunk + c) * rblocks + rb0 + i) * 64 + lane];
#pragma unroll
      for (int j = 0; j < CB; ++j) b[s][j] = Bf[(((size_t)s * nchunk + c) * cblocks + cb0 + j) * 64 + lane];
    }
#pragma unroll
    for (int g = 0; g < NS; ++g)
#pragma unroll
      for (int s = 0; s <= g; ++s)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[g][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s][i], b[g - s][j], acc[g][i][j], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const double cscale = cs[(size_t)(cb0 + j) * 64 + lane];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = 0.0;
#pragma unroll
        for (int g = NS - 1; g >= 0; --g) v += (double)acc[g][i][j][r] * __builtin_ldexp(1.0, -12 - 7 * g);
        v *= rs[((size_t)(rb0 + i) * 64 + lane) * 4 + r] * cscale;
        Cf[(((size_t)(rb0 + i) * cblocks + cb0 + j) * 64 + lane) * 4 + r] = v;
        ss += v * v;
      }
    // the four lanes that hold the same column (lane & 15): one partial per wave
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16) atomicAdd(&colss[16 * (cb0 + j) + lane], ss);
  }
}

// (Tried and removed: a software-pipelined form of the one-wave-per-SIMD kernel -- slice-major matrix instructions, the next chunk's slice s loaded
// into slice s's registers as soon as step s is over, B double-buffered.  The A / B operands of a matrix instruction must sit in the 256
// architectural VGPRs (only accumulators live in the other 256): 112 + 2 x 56 operand registers plus the loads in flight do not fit, the
// compiler shuttles them through accumulator registers and scratch, 0.625 ms.  A tile that can be double-buffered -- 2 x 2 blocks -- is
// L2-bound instead: the way on is LDS, two buffers.)
// The same contraction with the operand fragments staged ONCE per workgroup through LDS by global_load_lds_dwordx4 (memory -> LDS without a
// register in between: a wave instruction moves one whole 1-KiB fragment, lane l's 16 bytes to base + 16 l): 8 waves (4 x 2) of 2 x 2 blocks
// = a 128 x 64 piece; per k chunk 56 + 28 fragments = 84 KiB of LDS (one buffer: the next chunk's loads are issued as soon as every wave has
// its operands in registers, and run underneath the 112 matrix instructions of the chunk).
__global__ __launch_bounds__(512, 1) void ozaki_tile_lds(const v4i* __restrict__ Af, const v4i* __restrict__ Bf, int nchunk, int rblocks, int cblocks,
                                                        const double* __restrict__ rs, const double* __restrict__ cs, double* __restrict__ Cf,
                                                        double* __restrict__ colss) {
  constexpr int RB = 2, CB = 2, WR = 4, WC = 2;
  __shared__ v4i lA[NS][WR * RB][64];
  __shared__ v4i lB[NS][WC * CB][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv % WR, wc = wv / WR;
  const int rbw = blockIdx.x * WR * RB, cbw = blockIdx.y * WC * CB;       // the workgroup's first row / column block
  const int rb0 = rbw + wr * RB, cb0 = cbw + wc * CB;
  // the 84 fragments of a chunk dealt round the 8 waves: fragment f < 56 is A (slice f / 8, row block f % 8), else B (slice (f - 56) / 4, column block (f - 56) % 4)
  auto stage = [&](int c) {
#pragma unroll
    for (int u = 0; u < 11; ++u) {
      const int f = wv + 8 * u;
      if (f < 56) {
        const int sl = f >> 3, blk = f & 7;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Af + (((size_t)sl * nchunk + c) * rblocks + rbw + blk) * 64 + lane),
                                         (__attribute__((address_space(3))) void*)&lA[sl][blk][0], 16, 0, 0);
      } else if (f < 84) {
        const int sl = (f - 56) >> 2, blk = (f - 56) & 3;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bf + (((size_t)sl * nchunk + c) * cblocks + cbw + blk) * 64 + lane),
                                         (__attribute__((address_space(3))) void*)&lB[sl][blk][0], 16, 0, 0);
      }
    }
  };
  v4i acc[NS][RB][CB];
#pragma unroll
  for (int g = 0; g < NS; ++g)
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[g][i][j] = (v4i){0, 0, 0, 0};
  stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nchunk; ++c) {
    v4i a[NS][RB], b[NS][CB];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < RB; ++i) a[s][i] = lA[s][wr * RB + i][lane];
#pragma unroll
      for (int j = 0; j < CB; ++j) b[s][j] = lB[s][wc * CB + j][lane];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                        // every wave has its operands: the buffer is free
    if (c + 1 < nchunk) stage(c + 1);
#pragma unroll
    for (int g = 0; g < NS; ++g)
#pragma unroll
      for (int s = 0; s <= g; ++s)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[g][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s][i], b[g - s][j], acc[g][i][j], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // the next chunk is in LDS
  }
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const double cscale = cs[(size_t)(cb0 + j) * 64 + lane];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = 0.0;
#pragma unroll
        for (int g = NS - 1; g >= 0; --g) v += (double)acc[g][i][j][r] * __builtin_ldexp(1.0, -12 - 7 * g);
        v *= rs[((size_t)(rb0 + i) * 64 + lane) * 4 + r] * cscale;
        Cf[(((size_t)(rb0 + i) * cblocks + cb0 + j) * 64 + lane) * 4 + r] = v;
        ss += v * v;
      }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16) atomicAdd(&colss[16 * (cb0 + j) + lane], ss);
  }
}

// Two LDS buffers: a 64 x 64 piece per workgroup (4 waves of 2 x 2 blocks, one per SIMD), 28 + 28 fragments = 56 KiB per chunk and buffer.  The
// next chunk streams into the other buffer (global_load_lds) while this one is read -- the LDS reads are asynchronous too, so the matrix
// instructions of the first slices start while the later slices are still on their way -- and ONE barrier per chunk closes both.
__global__ __launch_bounds__(256, 1) void ozaki_tile_lds2(const v4i* __restrict__ Af, const v4i* __restrict__ Bf, int nchunk, int rblocks, int cblocks,
                                                         const double* __restrict__ rs, const double* __restrict__ cs, double* __restrict__ Cf,
                                                         double* __restrict__ colss) {
  constexpr int RB = 2, CB = 2;
  __shared__ v4i L[2][2][NS][4][64];                        // [buffer][A / B][slice][block][lane]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv & 1, wc = wv >> 1;
  const int rbw = blockIdx.x * 4, cbw = blockIdx.y * 4;
  const int rb0 = rbw + wr * RB, cb0 = cbw + wc * CB;
  // the 56 fragments of a chunk dealt round the 4 waves: f < 28 is A (slice f / 4, row block f % 4), else B -- all of it scalar arithmetic,
  // only the lane's 16-byte offset is a vector (branches and per-lane 64-bit addresses here cost 6x the whole kernel)
  auto stage = [&](int c, int buf) {
#pragma unroll
    for (int u = 0; u < 14; ++u) {
      const int f = wv + 4 * u;
      const int isB = f >= 28 ? 1 : 0, sl = (f - 28 * isB) >> 2, blk = f & 3;
      const size_t fi = isB ? ((size_t)sl * nchunk + c) * cblocks + cbw + blk : ((size_t)sl * nchunk + c) * rblocks + rbw + blk;
      const char* base = reinterpret_cast<const char*>(isB ? Bf : Af) + fi * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + lane * 16),
                                       (__attribute__((address_space(3))) void*)&L[buf][isB][sl][blk][0], 16, 0, 0);
    }
  };
  v4i acc[NS][RB][CB];
#pragma unroll
  for (int g = 0; g < NS; ++g)
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[g][i][j] = (v4i){0, 0, 0, 0};
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  auto chunk = [&](int c, int buf) {
    if (c + 1 < nchunk) stage(c + 1, buf ^ 1);
    v4i a[NS][RB], b[NS][CB];
#pragma unroll
    for (int s = 0; s < NS; ++s) {                          // (b first: step 0 needs every slice of B)
#pragma unroll
      for (int j = 0; j < CB; ++j) b[s][j] = L[buf][1][s][wc * CB + j][lane];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < RB; ++i) a[s][i] = L[buf][0][s][wr * RB + i][lane];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int t = NS - 1 - s; t >= 0; --t)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[s + t][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s][i], b[t][j], acc[s + t][i][j], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // this buffer has been read by every wave, the other one is full
  };
#pragma unroll 1
  for (int c = 0; c < nchunk; c += 2) {
    chunk(c, 0);
    if (c + 1 < nchunk) chunk(c + 1, 1);
  }
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const double cscale = cs[(size_t)(cb0 + j) * 64 + lane];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = 0.0;
#pragma unroll
        for (int g = NS - 1; g >= 0; --g) v += (double)acc[g][i][j][r] * __builtin_ldexp(1.0, -12 - 7 * g);
        v *= rs[((size_t)(rb0 + i) * 64 + lane) * 4 + r] * cscale;
        Cf[(((size_t)(rb0 + i) * cblocks + cb0 + j) * 64 + lane) * 4 + r] = v;
        ss += v * v;
      }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16) atomicAdd(&colss[16 * (cb0 + j) + lane], ss);
  }
}

// A RING of LDS slots, one slice of one chunk each (4 + 4 fragments = 8 KiB for a 64 x 64 piece), 16 slots = 2.3 chunks of look-ahead: the
// memory -> LDS stream never stops (global_load_lds, two fragments per wave and step), a slot is read into registers one step before its matrix
// instructions and refilled one step after.  Step u of a chunk brings slice u of A and of B and runs the pairs (s, t) with max(s, t) = u,
// s + t <= 6 (1, 3, 5, 7, 6, 4, 2 pairs): every pair exactly once, operands of earlier steps stay in registers.  One barrier per step; the
// loads complete in order, so "slot q is in" is s_waitcnt vmcnt(2 x (slots issued after it)).
#define RING 16
__global__ __launch_bounds__(256, 1) void ozaki_tile_ring(const v4i* __restrict__ Af, const v4i* __restrict__ Bf, int nchunk, int rblocks, int cblocks,
                                                         const double* __restrict__ rs, const double* __restrict__ cs, double* __restrict__ Cf,
                                                         double* __restrict__ colss) {
  constexpr int RB = 2, CB = 2;
  __shared__ v4i L[RING][2][4][64];                         // [slot][A / B][block][lane]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv & 1, wc = wv >> 1;
  const int rbw = blockIdx.x * 4, cbw = blockIdx.y * 4;
  const int rb0 = rbw + wr * RB, cb0 = cbw + wc * CB;
  const int nstep = nchunk * NS;
  // slot q = (chunk q / 7, slice q % 7): 8 fragments, wave w brings fragments 2 w and 2 w + 1 (0..3: A blocks, 4..7: B blocks); past the end
  // the last slice is fetched again (into a slot nobody reads any more) so that the count of loads in flight stays what the waits assume
  auto fill = [&](int q) {
    const int qq = q < nstep ? q : nstep - 1;
    const int c = qq / NS, sl = qq - c * NS, slot = q % RING;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = 2 * wv + h, isB = f >> 2, blk = f & 3;
      const size_t fi = isB ? ((size_t)sl * nchunk + c) * cblocks + cbw + blk : ((size_t)sl * nchunk + c) * rblocks + rbw + blk;
      const char* base = reinterpret_cast<const char*>(isB ? Bf : Af) + fi * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + lane * 16),
                                       (__attribute__((address_space(3))) void*)&L[slot][isB][blk][0], 16, 0, 0);
    }
  };
  v4i acc[NS][RB][CB];
#pragma unroll
  for (int g = 0; g < NS; ++g)
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[g][i][j] = (v4i){0, 0, 0, 0};
  v4i a[NS][RB], b[NS][CB];
  auto fetch = [&](int q, int u) {                          // slot q -> the registers of slice u
    const int slot = q % RING;
#pragma unroll
    for (int i = 0; i < RB; ++i) a[u][i] = L[slot][0][wr * RB + i][lane];
#pragma unroll
    for (int j = 0; j < CB; ++j) b[u][j] = L[slot][1][wc * CB + j][lane];
  };
  // prologue: slots 0 .. RING - 2 on their way, slot 0 in registers
  for (int q = 0; q < RING - 1; ++q) fill(q);
  // (NOT __syncthreads(): its fence would wait for every load in flight; and wait + barrier in ONE asm with a memory clobber: the compiler must
  //  not move the LDS reads of the slot above them)
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(2 * (RING - 2)) : "memory");
  fetch(0, 0);
#pragma unroll 1
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int q = c * NS + u;
      // slot q + 1 is in (every wave's part of it): behind the barrier it is read, and slot q - 1 -- read by everyone one step ago -- refilled
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(2 * (RING - 3)) : "memory");
      fill(q + RING - 1);
      // (the registers of slice (u + 1) % 7 were last used by the matrix instructions of the previous chunk's steps: program order)
      v4i an[RB], bn[CB];
      {
        const int slot = (q + 1) % RING;
#pragma unroll
        for (int i = 0; i < RB; ++i) an[i] = L[slot][0][wr * RB + i][lane];
#pragma unroll
        for (int j = 0; j < CB; ++j) bn[j] = L[slot][1][wc * CB + j][lane];
      }
      // the pairs of step u: (u, t), t = 0 .. min(u, 6 - u), and (s, u), s = 0 .. min(u - 1, 6 - u)
#pragma unroll
      for (int t = 0; t <= (u < NS - 1 - u ? u : NS - 1 - u); ++t)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[u + t][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u][i], b[t][j], acc[u + t][i][j], 0, 0, 0);
#pragma unroll
      for (int sl = 0; sl <= (u - 1 < NS - 1 - u ? u - 1 : NS - 1 - u); ++sl)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[sl + u][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[sl][i], b[u][j], acc[sl + u][i][j], 0, 0, 0);
      // hand the next slice's operands over (slice (u + 1) % 7 of this or the next chunk)
#pragma unroll
      for (int i = 0; i < RB; ++i) a[(u + 1) % NS][i] = an[i];
#pragma unroll
      for (int j = 0; j < CB; ++j) b[(u + 1) % NS][j] = bn[j];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const double cscale = cs[(size_t)(cb0 + j) * 64 + lane];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = 0.0;
#pragma unroll
        for (int g = NS - 1; g >= 0; --g) v += (double)acc[g][i][j][r] * __builtin_ldexp(1.0, -12 - 7 * g);
        v *= rs[((size_t)(rb0 + i) * 64 + lane) * 4 + r] * cscale;
        Cf[(((size_t)(rb0 + i) * cblocks + cb0 + j) * 64 + lane) * 4 + r] = v;
        ss += v * v;
      }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16) atomicAdd(&colss[16 * (cb0 + j) + lane], ss);
  }
}

// Two LDS buffers filled THROUGH REGISTERS (buffer loads two chunks ahead into staging registers, ds_write_b128 one chunk ahead), one barrier
// per chunk; the general form: NSL slices (the first NSL of the seven: a coarser split), a workgroup of WR x WC waves, each RB x CB blocks,
// i.e. a piece of 16 WR RB rows x 16 WC CB columns; LDS 2 x NSL x (WR RB + WC CB) KiB.
template <int NSL, int RB, int CB, int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC, 1) void ozaki_tile_reg2(const v4i* __restrict__ Af, const v4i* __restrict__ Bf, int nchunk, int rblocks, int cblocks,
                                                                  const double* __restrict__ rs, const double* __restrict__ cs, double* __restrict__ Cf,
                                                                  double* __restrict__ colss) {
  constexpr int NW = WR * WC, TR = WR * RB, TC = WC * CB;
  __shared__ v4i L[2][NSL][TR + TC][64];                    // [buffer][slice][A blocks | B blocks][lane]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv % WR, wc = wv / WR;
  const int rbw = blockIdx.x * TR, cbw = blockIdx.y * TC;
  const int rb0 = rbw + wr * RB, cb0 = cbw + wc * CB;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<v4i*>(Af), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<v4i*>(Bf), 0, -1, 0x00020000);
  const unsigned loff = lane * 16;
  // fragment f of a chunk (dealt round the waves, f = wv + NW u): slice f / (TR + TC), entry e = f % (TR + TC): A block e or B block e - TR
  // (dealt slice by slice instead -- entry wv + NW u of every slice, no division -- half the waves carry twice the loads: 0.153 against 0.139 ms)
  constexpr int NF = NSL * (TR + TC), PER = (NF + NW - 1) / NW;
  v4i stg[PER];
  auto fetch = [&](int c) {
    const int cc = c < nchunk ? c : nchunk - 1;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int f = wv + NW * u;
      if (f < NF) {
        const int sl = f / (TR + TC), e = f - sl * (TR + TC), isB = e >= TR ? 1 : 0;
        const unsigned fi = isB ? ((unsigned)sl * nchunk + cc) * cblocks + cbw + (e - TR) : ((unsigned)sl * nchunk + cc) * rblocks + rbw + e;
        stg[u] = __builtin_bit_cast(v4i, isB ? __builtin_amdgcn_raw_buffer_load_b128(rB, loff, (int)(fi * 1024u), 0)
                                             : __builtin_amdgcn_raw_buffer_load_b128(rA, loff, (int)(fi * 1024u), 0));
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
  v4i acc[NSL][RB][CB];
#pragma unroll
  for (int g = 0; g < NSL; ++g)
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[g][i][j] = (v4i){0, 0, 0, 0};
  fetch(0);
  put(0);
  fetch(1);
  __syncthreads();
  auto chunk = [&](int c, int buf) {
    // buffer buf ^ 1 was read in chunk c - 1 (the barrier at its end): chunk c + 1 goes in, chunk c + 2 into the staging registers
    put(buf ^ 1);
    fetch(c + 2);
    v4i a[NSL][RB], b[NSL][CB];
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
#pragma unroll
      for (int j = 0; j < CB; ++j) b[s][j] = L[buf][s][TR + wc * CB + j][lane];
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
          for (int j = 0; j < CB; ++j) acc[g][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s][i], b[g - s][j], acc[g][i][j], 0, 0, 0);
    __syncthreads();
  };
#pragma unroll 1
  for (int c = 0; c < nchunk; c += 2) {
    chunk(c, 0);
    if (c + 1 < nchunk) chunk(c + 1, 1);
  }
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const double cscale = cs[(size_t)(cb0 + j) * 64 + lane];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = 0.0;
#pragma unroll
        for (int g = NSL - 1; g >= 0; --g) v += (double)acc[g][i][j][r] * __builtin_ldexp(1.0, -12 - 7 * g);
        v *= rs[((size_t)(rb0 + i) * 64 + lane) * 4 + r] * cscale;
        Cf[(((size_t)(rb0 + i) * cblocks + cb0 + j) * 64 + lane) * 4 + r] = v;
        ss += v * v;
      }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16) atomicAdd(&colss[16 * (cb0 + j) + lane], ss);
  }
}

// the same register tile in fp64 (v_mfma_f64_16x16x4, operands straight from memory too): what THIS loop structure gives the fp64 pipe
typedef double v4d __attribute__((ext_vector_type(4)));
#define RB 4
#define CB 2
__global__ __launch_bounds__(256, 1) void fp64_tile(const double* __restrict__ A, const double* __restrict__ B, int K, int M, int N, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r0 = ((blockIdx.x * 2 + (wv & 1)) * RB) * 16, c0 = ((blockIdx.y * 2 + (wv >> 1)) * CB) * 16;
  v4d acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  const int c15 = lane & 15, q = lane >> 4;
#pragma unroll 4
  for (int k = 0; k < K; k += 4) {
    double a[RB], b[CB];
#pragma unroll
    for (int i = 0; i < RB; ++i) a[i] = A[(size_t)(k + q) * M + r0 + 16 * i + c15];
#pragma unroll
    for (int j = 0; j < CB; ++j) b[j] = B[(size_t)(k + q) * N + c0 + 16 * j + c15];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 1.2345e300) out[0] = s;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
  srand(12345);
  // ---------------- 1. operand layout
  std::vector<int8_t> ha(64 * 16), hb(64 * 16);
  for (auto& x : ha) x = (int8_t)(rand() % 127 - 63);
  for (auto& x : hb) x = (int8_t)(rand() % 127 - 63);
  v4i *da, *db, *dd;
  CHK(hipMalloc(&da, 1024)); CHK(hipMalloc(&db, 1024)); CHK(hipMalloc(&dd, 1024));
  CHK(hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice));
  CHK(hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dd);
  std::vector<int> hd(256);
  CHK(hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost));
  int H = -1, O = -1;
  for (int h = 0; h < 2 && H < 0; ++h)
    for (int o = 0; o < 2 && H < 0; ++o) {
      long ref[16][16];
      memset(ref, 0, sizeof(ref));
      for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb)
          for (int ta = 0; ta < 16; ++ta)
            for (int tb = 0; tb < 16; ++tb)
              if (kmap(h, la >> 4, ta) == kmap(h, lb >> 4, tb)) ref[la & 15][lb & 15] += (long)ha[la * 16 + ta] * hb[lb * 16 + tb];
      bool ok = true;
      for (int l = 0; l < 64 && ok; ++l)
        for (int r = 0; r < 4; ++r)
          if (hd[l * 4 + r] != ref[rmap(o, l >> 4, r)][l & 15]) { ok = false; break; }
      if (ok) { H = h; O = o; }
    }
  if (H < 0) { printf("operand layout of v_mfma_i32_16x16x64_i8 not recognised\n"); return 1; }
  printf("v_mfma_i32_16x16x64_i8: byte t of lane l is k = %s of row l %% 16; element r of lane l is row %s, column l %% 16\n",
         H == 0 ? "16 (l / 16) + t" : "8 (l / 16) + t %% 8 + 32 (t / 8)", O == 0 ? "4 (l / 16) + r" : "4 r + l / 16");
  // ---------------- 2. operands: a slab of a triangular inverse factor (wide dynamic range) and of kernel values in (0, 1]
  // default: 16 x 16 workgroups of 128 x 64, one per compute unit; ./ozaki_tile_probe.bin M N for a larger problem (multiples of 128)
  const int K = 4096, M = argc > 2 ? atoi(argv[1]) : 2048, N = argc > 2 ? atoi(argv[2]) : 1024;
  std::vector<double> A((size_t)K * M), B((size_t)K * N);
  for (size_t i = 0; i < A.size(); ++i) A[i] = ((rand() / (double)RAND_MAX) - 0.5) * ldexp(1.0, -(rand() % 20));
  for (size_t i = 0; i < B.size(); ++i) B[i] = exp(-8.0 * (rand() / (double)RAND_MAX));
  // ---------------- 3. slices, fragment by fragment
  const int nchunk = K / KC, rblocks = M / 16, cblocks = N / 16;
  auto slice = [&](const std::vector<double>& X, int cols, std::vector<int8_t>& F, std::vector<int>& expo) {
    F.assign((size_t)NS * nchunk * (cols / 16) * 1024, 0);
    expo.assign(cols, 0);
    for (int m = 0; m < cols; ++m) {
      double mx = 0.0;
      for (int k = 0; k < K; ++k) mx = fmax(mx, fabs(X[(size_t)k * cols + m]));
      expo[m] = mx > 0.0 ? ilogb(mx) + 1 : 0;
    }
    int qof[KC], tof[KC];                                    // which lane group / byte holds k
    for (int qq = 0; qq < 4; ++qq)
      for (int tt = 0; tt < 16; ++tt) { qof[kmap(H, qq, tt)] = qq; tof[kmap(H, qq, tt)] = tt; }
    for (int k = 0; k < K; ++k)
      for (int m = 0; m < cols; ++m) {
        double x = ldexp(X[(size_t)k * cols + m], 6 - expo[m]);        // |x| < 64
        const int c = k / KC, kk = k % KC, blk = m / 16, row = m % 16;
        const int q = qof[kk], t = tof[kk];
        for (int s = 0; s < NS; ++s) {
          const double d = nearbyint(x);
          F[((((size_t)s * nchunk + c) * (cols / 16) + blk) * 64 + 16 * q + row) * 16 + t] = (int8_t)d;
          x = (x - d) * 128.0;
        }
      }
  };
  std::vector<int8_t> FA, FB;
  std::vector<int> eA, eB;
  slice(A, M, FA, eA);
  slice(B, N, FB, eB);
  std::vector<double> hrs((size_t)rblocks * 256), hcs((size_t)cblocks * 64);
  for (int rb = 0; rb < rblocks; ++rb)
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) hrs[((size_t)rb * 64 + l) * 4 + r] = ldexp(1.0, eA[16 * rb + rmap(O, l >> 4, r)]);
  for (int cb = 0; cb < cblocks; ++cb)
    for (int l = 0; l < 64; ++l) hcs[(size_t)cb * 64 + l] = ldexp(1.0, eB[16 * cb + (l & 15)]);
  v4i *dFA, *dFB;
  double *drs, *dcs, *dC, *dss, *dA, *dB;
  CHK(hipMalloc(&dFA, FA.size())); CHK(hipMalloc(&dFB, FB.size()));
  CHK(hipMalloc(&drs, hrs.size() * 8)); CHK(hipMalloc(&dcs, hcs.size() * 8));
  CHK(hipMalloc(&dC, (size_t)M * N * 8)); CHK(hipMalloc(&dss, (size_t)N * 8));
  CHK(hipMalloc(&dA, A.size() * 8)); CHK(hipMalloc(&dB, B.size() * 8));
  CHK(hipMemcpy(dFA, FA.data(), FA.size(), hipMemcpyHostToDevice));
  CHK(hipMemcpy(dFB, FB.data(), FB.size(), hipMemcpyHostToDevice));
  CHK(hipMemcpy(drs, hrs.data(), hrs.size() * 8, hipMemcpyHostToDevice));
  CHK(hipMemcpy(dcs, hcs.data(), hcs.size() * 8, hipMemcpyHostToDevice));
  CHK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  CHK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
  // ---------------- 4. time
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const double flop = 2.0 * M * N * K, i8ops = 28.0 * flop;
  const double t_prod = flop / (0.93 * 78.6e12) * 1e3;       // what the product's fp64 kernel needs for this much work on the whole chip
  const double tiles = (double)M * N / (256.0 * 128.0);     // 256 x 128 tiles in the problem, spread over all compute units
  printf("tile kernels on %d x %d x %d, 7 slices / 28 products; the product's fp64 kernel needs %.3f ms for it (0.93 of 78.6 TFLOP/s) = %.3f ms per 256 x 128 x 4096 tile and compute unit\n",
         M, N, K, t_prod, t_prod * prop.multiProcessorCount / tiles);
  auto time_it = [&](auto launch, const char* what) -> int {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CHK(hipMemset(dss, 0, (size_t)N * 8));
      CHK(hipEventRecord(e0, 0));
      launch();
      CHK(hipEventRecord(e1, 0));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    printf("  %-64s %.3f ms = %.0f TOP/s int8 = %.2f of the 4.9 POP/s the matrix pipe sustains; %.2f x the fp64 kernel; %.3f ms per tile and compute unit\n", what, best,
           i8ops / best * 1e-9, i8ops / best * 1e-9 / 4900.0, best / t_prod, best * prop.multiProcessorCount / tiles);
    return 0;
  };
  if (time_it([&] { hipLaunchKernelGGL((ozaki_tile<2, 2, 2, 4>), dim3(M / 64, N / 128), dim3(512), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "8 waves of 2 x 2 blocks (two per SIMD, 256 registers each)")) return 1;
  if (time_it([&] { hipLaunchKernelGGL((ozaki_tile<4, 2, 2, 2>), dim3(M / 128, N / 64), dim3(256), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "4 waves of 4 x 2 blocks (one per SIMD, 512 registers each)")) return 1;
  if (time_it([&] { hipLaunchKernelGGL(ozaki_tile_lds, dim3(M / 128, N / 64), dim3(512), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "8 waves of 2 x 2 blocks, operands through LDS (global_load_lds)")) return 1;
  if (time_it([&] { hipLaunchKernelGGL(ozaki_tile_lds2, dim3(M / 64, N / 64), dim3(256), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "4 waves of 2 x 2 blocks, two LDS buffers, one barrier per chunk")) return 1;
  if (time_it([&] { hipLaunchKernelGGL(ozaki_tile_ring, dim3(M / 64, N / 64), dim3(256), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "4 waves of 2 x 2 blocks, ring of 16 one-slice LDS slots")) return 1;
  if (time_it([&] { hipLaunchKernelGGL((ozaki_tile_reg2<7, 2, 2, 2, 2>), dim3(M / 64, N / 64), dim3(256), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "4 waves of 2 x 2 blocks, two LDS buffers filled through registers")) return 1;
  if (time_it([&] { hipLaunchKernelGGL((ozaki_tile_reg2<7, 1, 2, 4, 2>), dim3(M / 64, N / 64), dim3(512), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "8 waves of 1 x 2 blocks (two per SIMD), two LDS buffers through registers")) return 1;
  // six slices (21 products; the time columns still count 28 products' worth of operations, so "x the fp64 kernel" is what matters)
  if (time_it([&] { hipLaunchKernelGGL((ozaki_tile_reg2<6, 1, 2, 4, 2>), dim3(M / 64, N / 64), dim3(512), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "SIX slices: 8 waves of 1 x 2 blocks, 64 x 64 piece")) return 1;
  if (time_it([&] { hipLaunchKernelGGL((ozaki_tile_reg2<6, 2, 2, 4, 2>), dim3(M / 128, N / 64), dim3(512), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "SIX slices: 8 waves of 2 x 2 blocks, 128 x 64 piece (144 KiB of LDS)")) return 1;
  if (time_it([&] { hipLaunchKernelGGL((ozaki_tile_reg2<6, 2, 2, 2, 4>), dim3(M / 64, N / 128), dim3(512), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "SIX slices: 8 waves of 2 x 2 blocks, 64 x 128 piece")) return 1;
  if (time_it([&] { hipLaunchKernelGGL((ozaki_tile_reg2<5, 2, 2, 4, 2>), dim3(M / 128, N / 64), dim3(512), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss); },
              "FIVE slices (15 products): 8 waves of 2 x 2 blocks, 128 x 64 piece")) return 1;
  {
    float best64 = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CHK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(fp64_tile, dim3(M / 128, N / 64), dim3(256), 0, 0, dA, dB, K, M, N, dC + 1);
      CHK(hipEventRecord(e1, 0));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best64) best64 = ms;
    }
    printf("  the same 4 x 2 register tile in fp64 (v_mfma_f64_16x16x4, operands straight from memory): %.3f ms = %.1f TFLOP/s (%.2f of peak) -- this loop structure is not the product's\n",
           best64, flop / best64 * 1e-9, flop / best64 * 1e-9 / 78.6);
  }
  CHK(hipMemset(dss, 0, (size_t)N * 8));
  CHK(hipMemset(dC, 0, (size_t)M * N * 8));
  const bool five = getenv("OZAKI_FIVE") != nullptr;
  const bool six = getenv("OZAKI_SIX") != nullptr;     // (further runs of the binary: the accuracy of the six- / five-slice kernels)
  if (five) hipLaunchKernelGGL((ozaki_tile_reg2<5, 2, 2, 4, 2>), dim3(M / 128, N / 64), dim3(512), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss);
  else if (six) hipLaunchKernelGGL((ozaki_tile_reg2<6, 2, 2, 4, 2>), dim3(M / 128, N / 64), dim3(512), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss);
  else hipLaunchKernelGGL((ozaki_tile_reg2<7, 1, 2, 4, 2>), dim3(M / 64, N / 64), dim3(512), 0, 0, dFA, dFB, nchunk, rblocks, cblocks, drs, dcs, dC, dss);
  printf("checked below: the %s-slice kernel\n", five ? "five" : (six ? "six" : "seven"));
  CHK(hipDeviceSynchronize());
  // ---------------- 5. accuracy: one 128 x 64 piece against long double, and its columns' sums of squares
  std::vector<double> hC((size_t)M * N), hss(N);
  CHK(hipMemcpy(hC.data(), dC, hC.size() * 8, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(hss.data(), dss, hss.size() * 8, hipMemcpyDeviceToHost));
  double emax = 0.0, e64max = 0.0, cmax = 0.0;
  const int m0 = 640, n0 = 320;
  for (int m = m0; m < m0 + 128; ++m)
    for (int n = n0; n < n0 + 64; ++n) {
      long double t = 0.0L;
      double t64 = 0.0;
      for (int k = 0; k < K; ++k) {
        t += (long double)A[(size_t)k * M + m] * (long double)B[(size_t)k * N + n];
        t64 = fma(A[(size_t)k * M + m], B[(size_t)k * N + n], t64);
      }
      // where the kernel put (m, n): block (m / 16, n / 16), lane 16 q + n % 16, element r with rmap(O, q, r) = m % 16
      int q = -1, r = -1;
      for (int qq = 0; qq < 4 && q < 0; ++qq)
        for (int rr = 0; rr < 4; ++rr)
          if (rmap(O, qq, rr) == m % 16) { q = qq; r = rr; break; }
      const double v = hC[(((size_t)(m / 16) * cblocks + n / 16) * 64 + 16 * q + n % 16) * 4 + r];
      emax = fmax(emax, fabs((double)((long double)v - t)));
      e64max = fmax(e64max, fabs((double)((long double)t64 - t)));
      cmax = fmax(cmax, fabs((double)t));
    }
  printf("accuracy on a 128 x 64 piece (max |C| = %.3g): int8 Ozaki max abs error %.3g (%.3g relative to max |C|), plain fp64 fma loop %.3g (%.3g)\n", cmax, emax,
         emax / cmax, e64max, e64max / cmax);
  double ssmax = 0.0;
  for (int n = n0; n < n0 + 8; ++n) {
    long double t2 = 0.0L;
    for (int m = 0; m < M; ++m) {
      long double t = 0.0L;
      for (int k = 0; k < K; ++k) t += (long double)A[(size_t)k * M + m] * (long double)B[(size_t)k * N + n];
      t2 += t * t;
    }
    ssmax = fmax(ssmax, fabs((double)(((long double)hss[n] - t2) / t2)));
  }
  printf("fused sum of squares over the %d rows, 8 columns: max relative error %.3g\n", M, ssmax);
  return 0;
}
