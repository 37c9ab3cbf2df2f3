// Bit-equality of the range-check-free sqrt / reciprocal of the diagonal-block kernel (fit.hip) with sqrt() and 1.0 / x over 2^32 random
// doubles in [2^-60, 2^61): hipcc --offload-arch=gfx950 -O3 tools/sqrt_probe.hip -o /tmp/sq && /tmp/sq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
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
__device__ __forceinline__ double rcp_normal(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  const double e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(e, r, r);
}
__global__ void k(unsigned long long seed, long n, unsigned long long* bad, double* ex) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  unsigned long long cnt = 0;
  for (; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned long long z = seed + i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    // mantissa random, exponent in [2^-60, 2^60]
    const int e = (int)(z % 121) - 60;
    const double x = ldexp(1.0 + (double)(z >> 12) * (1.0 / 4503599627370496.0), e);
    const double a = sqrt(x), b = sqrt_pos_normal(x);
    if (a != b) { ++cnt; ex[0] = x; }
    const double c = 1.0 / a, dd = rcp_normal(a);
    if (c != dd) { ++cnt; ex[0] = -a; }
    const double c2 = 1.0 / x, d2 = rcp_normal(x);
    if (c2 != d2) { ++cnt; ex[0] = -x; }
  }
  if (cnt) atomicAdd(bad, cnt);
}
int main() {
  unsigned long long* bad; double* ex;
  hipMalloc(&bad, 8); hipMalloc(&ex, 8); hipMemset(bad, 0, 8);
  const long n = 1L << 32;
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, 12345ull, n, bad, ex);
  hipDeviceSynchronize();
  unsigned long long h; double hx; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hx, ex, 8, hipMemcpyDeviceToHost);
  printf("sqrt_pos_normal vs sqrt, rcp_normal vs 1.0 / x: %llu mismatches in %ld samples (last x %.17g)\n", h, n, hx);
  return 0;
}
