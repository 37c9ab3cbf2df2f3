// What bounds the K(X,X) build (build_train_kernel)?  hipcc --offload-arch=gfx950 -O3 tools/kbuild_probe.hip -o /tmp/kbuild_probe && /tmp/kbuild_probe
// Same store pattern (m matrices of Np x Np doubles, 64 x 512 tiles on/above the diagonal, 16-B stores, a workgroup writes 4-KiB row
// segments), with the kernel's arithmetic (d = 8 differences + FMAs + one fp64 exp per element) or without it, and two tile shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int COMPUTE, int ROWS, int TRI>
__global__ __launch_bounds__(256) void probe(const double* __restrict__ X, int Np, double* __restrict__ S) {
  const int j = blockIdx.z;
  const int r0 = blockIdx.y * ROWS;
  const int gc = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (TRI && blockIdx.x * 512 + 511 < r0) return;
  double xa[8], xb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    xa[q] = X[(long)gc * 8 + q];
    xb[q] = X[(long)(gc + 1) * 8 + q];
  }
  double* Sj = S + (long)j * Np * Np;
  for (int rr = 0; rr < ROWS; ++rr) {
    const int gr = r0 + rr;
    double v0 = gr, v1 = gc;
    if (COMPUTE) {
      double ra = 0.0, rb = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double xq = X[(long)gr * 8 + q];
        const double d0 = xq - xa[q], d1 = xq - xb[q];
        ra += d0 * d0;
        rb += d1 * d1;
      }
      v0 = exp(-0.5 * ra);
      v1 = exp(-0.5 * rb);
    }
    *reinterpret_cast<v2d*>(Sj + (long)gr * Np + gc) = (v2d){v0, v1};
  }
}

int main() {
  const int m = 4;
  for (int Np : {4096, 8192}) {
    double *X, *S;
    CHK(hipMalloc(&X, sizeof(double) * Np * 8));
    CHK(hipMalloc(&S, sizeof(double) * (size_t)Np * Np * m));
    CHK(hipMemset(X, 0, sizeof(double) * Np * 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
#define RUN(COMPUTE, ROWS, TRI, label)                                                                                      \
    {                                                                                                                       \
      dim3 grid(Np / 512, Np / ROWS, m);                                                                                    \
      hipLaunchKernelGGL((probe<COMPUTE, ROWS, TRI>), grid, dim3(256), 0, 0, X, Np, S);                                     \
      CHK(hipDeviceSynchronize());                                                                                          \
      float best = 1e30f;                                                                                                   \
      for (int r = 0; r < 5; ++r) {                                                                                         \
        CHK(hipEventRecord(e0, 0));                                                                                         \
        hipLaunchKernelGGL((probe<COMPUTE, ROWS, TRI>), grid, dim3(256), 0, 0, X, Np, S);                                   \
        CHK(hipEventRecord(e1, 0));                                                                                         \
        CHK(hipEventSynchronize(e1));                                                                                       \
        float ms;                                                                                                           \
        CHK(hipEventElapsedTime(&ms, e0, e1));                                                                              \
        best = ms < best ? ms : best;                                                                                       \
      }                                                                                                                     \
      double tiles = 0;                                                                                                     \
      for (int by = 0; by < Np / ROWS; ++by)                                                                                \
        for (int bx = 0; bx < Np / 512; ++bx) tiles += (TRI && bx * 512 + 511 < by * ROWS) ? 0 : 1;                         \
      const double bytes = tiles * ROWS * 512 * 8.0 * m;                                                                    \
      printf("Np %5d  %-34s %8.1f MB  %7.3f ms  %5.2f TB/s\n", Np, label, bytes / 1e6, best, bytes / best * 1e-9);           \
    }
    RUN(0, 64, 0, "full, no arithmetic, 64 x 512")
    RUN(0, 64, 1, "triangle, no arithmetic, 64 x 512")
    RUN(1, 64, 1, "triangle, exp, 64 x 512")
    RUN(1, 64, 0, "full, exp, 64 x 512")
    RUN(0, 16, 1, "triangle, no arithmetic, 16 x 512")
    RUN(1, 16, 1, "triangle, exp, 16 x 512")
    RUN(0, 128, 1, "triangle, no arithmetic, 128 x 512")
    RUN(1, 128, 1, "triangle, exp, 128 x 512")
    CHK(hipFree(X));
    CHK(hipFree(S));
  }
  return 0;
}
