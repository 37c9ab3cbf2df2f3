// On-box calibration of the fp64 MFMA ceiling (v_mfma_f64_16x16x4_f64) on MI355X.
// Pure register-resident MFMA loop, W waves per SIMD, NACC independent accumulators per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void peak(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; it += 16) {
#pragma unroll
    for (int rep = 0; rep < 16; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = ncu * blocks_per_cu;
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(peak<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1.0, 0.5);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * iters * NACC * 2048.0;
  printf("nacc=%2d waves/SIMD=%d  %8.3f ms  %7.2f TFLOP/s\n", NACC, blocks_per_cu, best, flops / best / 1e9);
  hipFree(out);
}

int main() {
  int ncu = 0, clk = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("CUs=%d clock=%d kHz  spec peak = CUs*4*2048/64*clk = %.2f TFLOP/s\n", ncu, clk, ncu * 4.0 * 2048 / 64 * clk * 1e3 / 1e12);
  run<4>(1, 20000);
  run<16>(1, 5000);
  run<16>(2, 5000);
  run<4>(2, 20000);
  return 0;
}
