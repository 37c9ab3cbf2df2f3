// Variants of the two HBM-write-bound kernels of the path (K(X,X) build, fit.hip; cross kernel K(X,X*), predict.hip) side by side, to see what
// bounds them: hipcc --offload-arch=gfx950 -O3 tools/hbm_kernel_probe.hip -o /tmp/hbm_probe && /tmp/hbm_probe
//   store-only (several shapes: the chip's plain store rate for THIS pattern), the library's arithmetic (ocml exp, padding / diagonal selects),
//   exp without its range handling (arguments are <= 0 and results never denormal in a kernel matrix), interior tiles without selects,
//   more columns per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// ocml's exp (double) without the two range selects: same operations on the same operands for every x in [-745, 0]
__device__ __forceinline__ double exp_nonpos(double x) {
  x = __builtin_fmax(x, -1100.0);
  const double dn = __builtin_rint(x * 0x1.71547652b82fep+0);
  const double t = __builtin_fma(-dn, 0x1.abc9e3b39803fp-56, __builtin_fma(-dn, 0x1.62e42fefa39efp-1, x));
  double p = __builtin_fma(t, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
  p = __builtin_fma(t, p, 0x1.71dee623fde64p-19);
  p = __builtin_fma(t, p, 0x1.a01997c89e6b0p-16);
  p = __builtin_fma(t, p, 0x1.a01a014761f6ep-13);
  p = __builtin_fma(t, p, 0x1.6c16c1852b7b0p-10);
  p = __builtin_fma(t, p, 0x1.1111111122322p-7);
  p = __builtin_fma(t, p, 0x1.55555555502a1p-5);
  p = __builtin_fma(t, p, 0x1.5555555555511p-3);
  p = __builtin_fma(t, p, 0x1.000000000000bp-1);
  p = __builtin_fma(t, p, 1.0);
  p = __builtin_fma(t, p, 1.0);
  return __builtin_ldexp(p, (int)dn);
}

// VAR: 0 store only; 1 library form (exp, selects); 2 exp_nonpos + selects; 3 exp_nonpos, no selects (interior tile)
template <int VAR, int ROWS>
__global__ __launch_bounds__(256) void kbuild(const double* __restrict__ X, int N, int Np, double* __restrict__ S, double variance, double dg) {
  __shared__ double xr[ROWS * 8];
  const int j = blockIdx.z;
  const int r0 = blockIdx.y * ROWS;
  const int gc = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (blockIdx.x * 512 + 511 < r0) return;
  for (int idx = threadIdx.x; idx < ROWS * 8; idx += 256) xr[idx] = X[(long)r0 * 8 + idx];
  double xa[8], xb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    xa[q] = X[(long)gc * 8 + q];
    xb[q] = X[(long)(gc + 1) * 8 + q];
  }
  __syncthreads();
  double* Sj = S + (long)j * Np * Np;
  const bool c0 = gc < N, c1 = gc + 1 < N;
#pragma unroll 1
  for (int rr = 0; rr < ROWS; rr += 4) {
    double v[4][2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int gr = r0 + rr + e;
      if (VAR == 0) {
        v[e][0] = gr;
        v[e][1] = gc;
        continue;
      }
      double ra = 0.0, rb = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double xq = xr[(rr + e) * 8 + q];
        const double d0 = xq - xa[q], d1 = xq - xb[q];
        ra += d0 * d0;
        rb += d1 * d1;
      }
      double k0, k1;
      if (VAR == 1) {
        k0 = variance * exp(-0.5 * ra);
        k1 = variance * exp(-0.5 * rb);
      } else {
        k0 = variance * exp_nonpos(-0.5 * ra);
        k1 = variance * exp_nonpos(-0.5 * rb);
      }
      if (VAR == 3) {
        v[e][0] = k0;
        v[e][1] = k1;
      } else {
        const bool rreal = gr < N;
        v[e][0] = rreal ? (gr == gc ? variance + dg : (c0 ? k0 : 0.0)) : (gr == gc ? 1.0 : 0.0);
        v[e][1] = rreal ? (gr == gc + 1 ? variance + dg : (c1 ? k1 : 0.0)) : (gr == gc + 1 ? 1.0 : 0.0);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<v2d*>(Sj + (long)(r0 + rr + e) * Np + gc) = (v2d){v[e][0], v[e][1]};
  }
}

// plain streaming store of the same bytes, one 16-B store per lane per iteration, grid-stride: the chip's store rate
template <int NT = 0>
__global__ __launch_bounds__(256) void fill(double* __restrict__ S, long n2) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
    const v2d v = (v2d){(double)i, 1.0};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(S) + i);
    else reinterpret_cast<v2d*>(S)[i] = v;
  }
}

// cross kernel: VAR 0 store only; 1 library form (exp + compensated mean); 2 exp_nonpos + compensated mean; 3 exp_nonpos, plain mean;
// COLS columns per thread (2 or 4)
__device__ __forceinline__ void dd_fma_acc(double& s, double& c, double a, double b) {
#pragma clang fp contract(off)
  const double p = a * b;
  const double e = __builtin_fma(a, b, -p);
  const double t = s + p;
  const double z = t - s;
  c += ((s - (t - z)) + (p - z)) + e;
  s = t;
}
template <int VAR, int COLS, int NT = 0>
__global__ __launch_bounds__(256) void cross(const double* __restrict__ X, int Np, const double* __restrict__ Xc, const double* __restrict__ al,
                                             double* __restrict__ K, long ldk, double* __restrict__ mp, int nsplit, double variance) {
  const int j = blockIdx.z, split = blockIdx.y;
  const int c = (blockIdx.x * 256 + threadIdx.x) * COLS;
  double xc[COLS][8];
#pragma unroll
  for (int u = 0; u < COLS; ++u)
#pragma unroll
    for (int q = 0; q < 8; ++q) xc[u][q] = Xc[(long)(c + u) * 8 + q];
  const int nblk = Np / 128, bps = (nblk + nsplit - 1) / nsplit, b0 = split * bps, b1 = b0 + bps > nblk ? nblk : b0 + bps;
  double* Kj = K + (long)j * Np * ldk;
  for (int blk = b0; blk < b1; ++blk) {
    double mean[COLS], lo[COLS];
#pragma unroll
    for (int u = 0; u < COLS; ++u) mean[u] = lo[u] = 0.0;
    for (int kk = blk * 128; kk < blk * 128 + 128; ++kk) {
      double v[COLS];
      if (VAR == 0) {
#pragma unroll
        for (int u = 0; u < COLS; ++u) v[u] = kk + u;
      } else {
        double r[COLS];
#pragma unroll
        for (int u = 0; u < COLS; ++u) r[u] = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const double xq = X[(long)kk * 8 + q];
#pragma unroll
          for (int u = 0; u < COLS; ++u) {
            const double d = xq - xc[u][q];
            r[u] += d * d;
          }
        }
        const double a = al[kk];
#pragma unroll
        for (int u = 0; u < COLS; ++u) {
          v[u] = variance * (VAR == 1 ? exp(-0.5 * r[u]) : exp_nonpos(-0.5 * r[u]));
          if (VAR == 3) mean[u] += v[u] * a;
          else dd_fma_acc(mean[u], lo[u], v[u], a);
        }
      }
#pragma unroll
      for (int u = 0; u < COLS; u += 2) {
        if (NT) __builtin_nontemporal_store((v2d){v[u], v[u + 1]}, reinterpret_cast<v2d*>(Kj + (long)kk * ldk + c + u));
        else *reinterpret_cast<v2d*>(Kj + (long)kk * ldk + c + u) = (v2d){v[u], v[u + 1]};
      }
    }
#pragma unroll
    for (int u = 0; u < COLS; ++u) mp[((long)blk * gridDim.z + j) * ldk + c + u] = mean[u] + lo[u];
  }
}

// exp_nonpos against exp: bit for bit on a sweep of [-745.2, 0] plus the neighbourhoods of the reduction's breakpoints
__global__ void exp_check(unsigned long long n, unsigned long long* mism, double* worst) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // low-discrepancy sweep of the range + a dense cluster near 0 (kernel values of nearby points)
  const double u = (double)((i * 0x9E3779B97F4A7C15ull) >> 11) * (1.0 / 9007199254740992.0);
  const double x = (i & 3) == 1 ? -745.2 * u : ((i & 3) == 3 ? -1e12 * u * u : -u * u * 1e-3 * (double)(i % 4096));
  const double a = exp(x), b = exp_nonpos(x);
  if (__double_as_longlong(a) != __double_as_longlong(b)) {
    atomicAdd(mism, 1ull);
    *worst = x;
  }
}

template <typename F>
static double timeit(F f, int reps = 20) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  f();
  f();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const int m = 4, Np = 4096, N = 4096, C = 65536;
  double *X, *S, *Xc, *al, *K, *mp;
  CHK(hipMalloc(&X, sizeof(double) * Np * 8));
  CHK(hipMalloc(&S, sizeof(double) * (size_t)Np * Np * m));
  CHK(hipMalloc(&Xc, sizeof(double) * C * 8));
  CHK(hipMalloc(&al, sizeof(double) * Np));
  CHK(hipMalloc(&K, sizeof(double) * (size_t)Np * C * m));
  CHK(hipMalloc(&mp, sizeof(double) * (size_t)(Np / 128) * m * C));
  {
    double* h = new double[(size_t)C * 8];
    for (size_t i = 0; i < (size_t)C * 8; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 / 0.6;
    CHK(hipMemcpy(X, h, sizeof(double) * Np * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(Xc, h, sizeof(double) * C * 8, hipMemcpyHostToDevice));
    for (int i = 0; i < Np; ++i) h[i] = (i % 7) - 3.0;
    CHK(hipMemcpy(al, h, sizeof(double) * Np, hipMemcpyHostToDevice));
    delete[] h;
  }
  // bytes of the K build: 64 x 512 tiles on / above the diagonal
  double kb_bytes = 0;
  for (int rb = 0; rb < Np / 64; ++rb)
    for (int cb = 0; cb < Np / 512; ++cb)
      if (cb * 512 + 511 >= rb * 64) kb_bytes += 64.0 * 512 * 8;
  kb_bytes *= m;
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    const long n2 = (long)(kb_bytes / 16);
    const double ms = timeit([&] { hipLaunchKernelGGL(fill<0>, dim3(blocks), dim3(256), 0, 0, S, n2); });
    printf("plain 16-B streaming stores of the K build's %.0f MB, %5d workgroups: %.3f ms = %.2f TB/s\n", kb_bytes / 1e6, blocks, ms, kb_bytes / ms / 1e9);
  }
  for (int blocks : {4096, 16384}) {
    const long n2 = (long)(kb_bytes / 16);
    const double ms = timeit([&] { hipLaunchKernelGGL(fill<1>, dim3(blocks), dim3(256), 0, 0, S, n2); });
    printf("NON-TEMPORAL 16-B streaming stores of the same bytes,       %5d workgroups: %.3f ms = %.2f TB/s\n", blocks, ms, kb_bytes / ms / 1e9);
  }
  const dim3 gk(Np / 512, Np / 64, m);
#define KB(V, what) { const double ms = timeit([&] { hipLaunchKernelGGL((kbuild<V, 64>), gk, dim3(256), 0, 0, X, N, Np, S, 1.0, 1e-6); }); \
    printf("K build %-58s %.3f ms = %.2f TB/s (%.2f of 8)\n", what, ms, kb_bytes / ms / 1e9, kb_bytes / ms / 1e9 / 8.0); }
  KB(0, "store only (its tiles, its 16-B stores)");
#define KBR(V, R, what) { const dim3 g2(Np / 512, Np / R, m); const double ms = timeit([&] { hipLaunchKernelGGL((kbuild<V, R>), g2, dim3(256), 0, 0, X, N, Np, S, 1.0, 1e-6); }); \
    printf("K build %-58s %.3f ms = %.2f TB/s (%.2f of 8)\n", what, ms, kb_bytes / ms / 1e9, kb_bytes / ms / 1e9 / 8.0); }
  KBR(0, 32, "store only, 32-row tiles");
  KBR(0, 16, "store only, 16-row tiles");
  KBR(0, 8, "store only, 8-row tiles");
  KBR(3, 32, "no range handling, no selects, 32-row tiles");
  KBR(3, 16, "no range handling, no selects, 16-row tiles");
  KBR(3, 8, "no range handling, no selects, 8-row tiles");
  KBR(2, 16, "no range handling, selects, 16-row tiles");
  KB(1, "library form (ocml exp, padding / diagonal selects)");
  KB(2, "exp without range handling, selects");
  KB(3, "exp without range handling, no selects (interior tile)");
  const double cr_bytes = 8.0 * Np * (double)C * m;
  const int nsplit = 1;
#define CR(V, COLS, what) { const dim3 gc(C / (256 * COLS), nsplit, m); const double ms = timeit([&] { hipLaunchKernelGGL((cross<V, COLS>), gc, dim3(256), 0, 0, X, Np, Xc, al, K, (long)C, mp, nsplit, 1.0); }, 5); \
    printf("cross %-60s %.3f ms = %.2f TB/s (%.2f of 8)\n", what, ms, cr_bytes / ms / 1e9, cr_bytes / ms / 1e9 / 8.0); }
  CR(0, 2, "store only, 2 columns per thread");
  CR(0, 4, "store only, 4 columns per thread");
#define CRN(V, COLS, what) { const dim3 gc(C / (256 * COLS), nsplit, m); const double ms = timeit([&] { hipLaunchKernelGGL((cross<V, COLS, 1>), gc, dim3(256), 0, 0, X, Np, Xc, al, K, (long)C, mp, nsplit, 1.0); }, 5); \
    printf("cross %-60s %.3f ms = %.2f TB/s (%.2f of 8)\n", what, ms, cr_bytes / ms / 1e9, cr_bytes / ms / 1e9 / 8.0); }
  CRN(0, 2, "store only, 2 columns per thread, NON-TEMPORAL stores");
  CRN(2, 2, "exp without range handling, compensated mean, 2 columns, NON-TEMPORAL stores");
  CR(1, 2, "library form (ocml exp, compensated mean), 2 columns");
  CR(2, 2, "exp without range handling, compensated mean, 2 columns");
  CR(3, 2, "exp without range handling, plain mean, 2 columns");
  CR(2, 4, "exp without range handling, compensated mean, 4 columns");
  {
    unsigned long long* mism; double* worst;
    CHK(hipMalloc(&mism, 8)); CHK(hipMalloc(&worst, 8));
    CHK(hipMemset(mism, 0, 8)); CHK(hipMemset(worst, 0, 8));
    const unsigned long long n = 1ull << 30;
    hipLaunchKernelGGL(exp_check, dim3((unsigned)(n / 256)), dim3(256), 0, 0, n, mism, worst);
    unsigned long long hm = 0; double hw = 0;
    CHK(hipMemcpy(&hm, mism, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hw, worst, 8, hipMemcpyDeviceToHost));
    printf("exp_nonpos vs exp on 2^30 arguments in [-1e12, 0] (half of them in [-745.2, 0]): %llu mismatching bit patterns (last at x = %.17g)\n", hm, hw);
  }
  return 0;
}
