// Sustained I8 MFMA rate of this MI355X (hardware half): a register-only loop of independent
// v_mfma_i32_16x16x64_i8 (gfx950), 8 accumulator tiles per wave, W waves per SIMD, every CU busy.  An error-free int8 split of the
// fp64 variance contraction (oracle/ozaki_numerics_probe.py) needs s (s + 1) / 2 int8 GEMMs per fp64 GEMM: what it could buy is this
// rate over the fp64 MFMA rate (tools/mfma_peak_probe.hip), divided by that count.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_i8_peak.hip -o /tmp/mfma_i8_peak && /tmp/mfma_i8_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int THREADS>
__global__ __launch_bounds__(THREADS, 1) void i8_loop(int* out, int iters, int seed) {
  v4i acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (v4i){0, 0, 0, 0};
  v4i a[2], b[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) a[i] = (v4i){seed + (int)threadIdx.x, seed + i, seed * 3, seed ^ i};
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = (v4i){seed - (int)threadIdx.x, seed - i, seed * 5, seed ^ (i + 7)};
  if (seed >= 1000) {                                       // dense operands: every byte a hashed value (what digits of real data look like)
    unsigned h = (unsigned)seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { h = h * 1664525u + 1013904223u; a[i][e] = (int)(h ^ (h >> 13)); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { h = h * 1664525u + 1013904223u; b[i][e] = (int)(h ^ (h >> 13)); }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 0x12345678) out[0] = s;
}

// the same with v_mfma_i32_32x32x32_i8 (65 536 operations, 8 passes): 4 accumulator tiles of 16 registers per wave
typedef int v16i __attribute__((ext_vector_type(16)));
template <int THREADS>
__global__ __launch_bounds__(THREADS, 1) void i8_loop32(int* out, int iters, int seed) {
  v16i acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0;
  v4i a[2], b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a[i] = (v4i){seed + (int)threadIdx.x, seed + i, seed * 3, seed ^ i};
#pragma unroll
  for (int i = 0; i < 2; ++i) b[i] = (v4i){seed - (int)threadIdx.x, seed - i, seed * 5, seed ^ (i + 7)};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 0x12345678) out[0] = s;
}

template <int THREADS>
static int run32(int cus, int* out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 200000;
  hipLaunchKernelGGL(i8_loop32<THREADS>, dim3(cus), dim3(THREADS), 0, 0, out, 100, 1);
  CHK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(i8_loop32<THREADS>, dim3(cus), dim3(THREADS), 0, 0, out, iters, 1);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double ops = (double)cus * (THREADS / 64) * iters * 4.0 * (2.0 * 32 * 32 * 32);
  printf("32x32x32: %4d threads per CU (%d waves per SIMD): %.3f ms = %.1f TOP/s  (%.2f cycles per MFMA and SIMD at 2.4 GHz)\n", THREADS, THREADS / 256, best,
         ops / best * 1e-9, best * 1e-3 * 2.4e9 / ((double)(THREADS / 256) * iters * 4.0));
  return 0;
}

template <int THREADS>
static int run(int cus, int* out, hipEvent_t e0, hipEvent_t e1, int seed = 1) {
  if (seed >= 1000) printf("dense operand bytes: ");
  for (int iters : {200000}) {
    hipLaunchKernelGGL(i8_loop<THREADS>, dim3(cus), dim3(THREADS), 0, 0, out, 100, 1);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      CHK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(i8_loop<THREADS>, dim3(cus), dim3(THREADS), 0, 0, out, iters, seed);
      CHK(hipEventRecord(e1, 0));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    const double ops = (double)cus * (THREADS / 64) * iters * 8.0 * (2.0 * 16 * 16 * 64);
    printf("%4d threads per CU (%d waves per SIMD), %6d iterations: %.3f ms = %.1f TOP/s  (%.2f cycles per MFMA and SIMD at 2.4 GHz)\n", THREADS,
           THREADS / 256, iters, best, ops / best * 1e-9, best * 1e-3 * 2.4e9 / ((double)(THREADS / 256) * iters * 8.0));
  }
  return 0;
}

int main() {
  int* out;
  CHK(hipMalloc(&out, 8));
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
  if (run<256>(prop.multiProcessorCount, out, e0, e1)) return 1;
  if (run<512>(prop.multiProcessorCount, out, e0, e1)) return 1;
  if (run<1024>(prop.multiProcessorCount, out, e0, e1)) return 1;
  if (run<512>(prop.multiProcessorCount, out, e0, e1, 1234)) return 1;
  if (run<1024>(prop.multiProcessorCount, out, e0, e1, 1234)) return 1;
  if (run32<256>(prop.multiProcessorCount, out, e0, e1)) return 1;
  if (run32<512>(prop.multiProcessorCount, out, e0, e1)) return 1;
  if (run32<1024>(prop.multiProcessorCount, out, e0, e1)) return 1;
  return 0;
}
