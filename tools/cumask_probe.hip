// Probe (timing / placement only, not product code): does hipExtStreamCreateWithCUMask confine a stream's workgroups to the
// masked compute units on this box, which (XCC, SE, CU) does mask bit i select, and what does a latency-bound single-workgroup
// kernel cost on reserved CUs while the other CUs run an MFMA-heavy kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/cumask_probe.hip -o tools/cumask_probe.bin && tools/cumask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_burn(double* out, int iters) {
  v4d acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-5 + 1.0;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// latency chain: 128 dependent steps of (LDS write, barrier, LDS read, sqrt, divide) like the diagonal-block kernel
__global__ __launch_bounds__(256) void chain_kernel(double* out, int steps) {
  __shared__ double buf[256];
  double v = 1.0 + threadIdx.x * 1e-9;
  for (int k = 0; k < steps; ++k) {
    if (threadIdx.x == (k & 255)) buf[k & 255] = sqrt(v + 2.0);
    __syncthreads();
    const double p = buf[k & 255];
    v = v / p + 1.0;
    __syncthreads();
  }
  out[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  printf("CUs %d\n", prop.multiProcessorCount);
  const int words = 8;   // 256 bits
  uint32_t lo[words] = {0}, hi[words];
  const int reserved = 8;
  for (int i = 0; i < reserved; ++i) lo[i / 32] |= 1u << (i % 32);
  for (int w = 0; w < words; ++w) hi[w] = ~lo[w];
  hipStream_t s_res, s_bulk, s_plain;
  CHK(hipExtStreamCreateWithCUMask(&s_res, words, lo));
  CHK(hipExtStreamCreateWithCUMask(&s_bulk, words, hi));
  CHK(hipStreamCreate(&s_plain));
  uint32_t got[words];
  CHK(hipExtStreamGetCUMask(s_res, words, got));
  printf("reserved-stream mask words: %08x %08x ...\n", got[0], got[1]);
  unsigned* d_out;
  double* d_buf;
  CHK(hipMalloc(&d_out, sizeof(unsigned) * 2 * 4096));
  CHK(hipMalloc(&d_buf, sizeof(double) * 256 * 8192));
  std::vector<unsigned> h(2 * 4096);
  auto census = [&](hipStream_t s, const char* name, int nblk) {
    hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, s, d_out, 20000);
    CHK(hipStreamSynchronize(s));
    CHK(hipMemcpy(h.data(), d_out, sizeof(unsigned) * 2 * nblk, hipMemcpyDeviceToHost));
    std::set<unsigned> cus;
    for (int b = 0; b < nblk; ++b) {
      const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
    }
    printf("%s: %d workgroups landed on %zu distinct (xcc,se,sh,cu):", name, nblk, cus.size());
    int n = 0;
    for (unsigned c : cus) { if (n++ < 16) printf(" %x.%x.%x.%x", c >> 12, (c >> 8) & 0xf, (c >> 4) & 0xf, c & 0xf); }
    printf("\n");
  };
  census(s_plain, "plain stream", 2048);
  census(s_res, "reserved (8-bit mask)", 256);
  census(s_bulk, "bulk (complement mask)", 2048);
  // latency chain alone vs beside an MFMA burn on the complement / on a plain stream
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  auto time_chain = [&](hipStream_t sc, hipStream_t sb, const char* name) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      if (sb) hipLaunchKernelGGL(mfma_burn, dim3(4096), dim3(256), 0, sb, d_buf, 4000);
      CHK(hipEventRecord(e0, sc));
      hipLaunchKernelGGL(chain_kernel, dim3(4), dim3(256), 0, sc, d_buf + 256 * 4096, 128);
      CHK(hipEventRecord(e1, sc));
      CHK(hipDeviceSynchronize());
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-58s chain of 128 steps x 4 WGs: %.1f us\n", name, best * 1e3f);
  };
  time_chain(s_plain, nullptr, "alone, plain stream");
  time_chain(s_res, nullptr, "alone, reserved CUs");
  time_chain(s_plain, s_bulk, "plain stream beside MFMA burn on the complement mask");
  time_chain(s_res, s_bulk, "reserved CUs beside MFMA burn on the complement mask");
  hipStream_t s_plain2;
  CHK(hipStreamCreate(&s_plain2));
  time_chain(s_plain, s_plain2, "plain stream beside MFMA burn on a plain stream");
  time_chain(s_res, s_plain2, "reserved CUs beside MFMA burn on a plain stream");
  // burn throughput with and without the 8 CUs
  for (hipStream_t s : {s_plain2, s_bulk}) {
    CHK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(mfma_burn, dim3(4096), dim3(256), 0, s, d_buf, 4000);
    CHK(hipEventRecord(e1, s));
    CHK(hipDeviceSynchronize());
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("mfma burn on %s: %.3f ms\n", s == s_bulk ? "complement mask (248 CUs)" : "plain stream (256 CUs)", ms);
  }
  return 0;
}
