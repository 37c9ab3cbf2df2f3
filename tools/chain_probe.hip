// Where do the ~370 ns per elimination step of the 128x128 diagonal-block kernel go?  The unblocked register-cyclic
// Cholesky sweep (potrf_diag_kernel's phase 1) with parts switched off (results are then wrong -- timing only):
//   bit 0: no IEEE sqrt / divide (a multiply instead)   bit 1: no workgroup barrier   bit 2: no LDS broadcast (registers)
//   bit 3: no trailing FMAs                               bit 4: no owner branch (every thread does the pivot work)
//   hipcc --offload-arch=gfx950 -O3 tools/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define NB 128

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(double* S, int reps) {
  __shared__ double rowbuf[2][NB];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  double* blk = S + (long)blockIdx.x * NB * NB;
  double a[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a[i][j] = blk[(ty + 16 * i) * NB + tx + 16 * j];
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
#pragma unroll 1
      for (int kk = 0; kk < 16; ++kk) {
        const int k = kb * 16 + kk;
        double* rb = rowbuf[k & 1];
        if ((MODE & 16) || ty == kk) {
          double piv = __shfl(a[kb][kb], (ty & 3) * 16 + kk, 64);
          if (!(piv > 0.0)) piv = 1.0;
          double ukk, inv;
          if (MODE & 1) { ukk = piv * 0.5; inv = piv * 0.25; }
          else { ukk = sqrt(piv); inv = 1.0 / ukk; }
#pragma unroll
          for (int j = kb; j < 8; ++j) {
            const int c = tx + 16 * j;
            const double v = a[kb][j] * inv;
            if (c > k) a[kb][j] = v;
            else if (c == k) a[kb][j] = ukk;
            if (!(MODE & 4)) rb[c] = (c > k) ? v : 0.0;
          }
        }
        if (!(MODE & 2)) __syncthreads();
        double ur[8], uc[8];
#pragma unroll
        for (int i = kb; i < 8; ++i) {
          if (MODE & 4) { ur[i] = a[i][kb] * 1e-3; uc[i] = a[kb][i] * 1e-3; }
          else { ur[i] = rb[ty + 16 * i]; uc[i] = rb[tx + 16 * i]; }
        }
        if (!(MODE & 8)) {
#pragma unroll
          for (int i = kb; i < 8; ++i)
#pragma unroll
            for (int j = kb; j < 8; ++j) a[i][j] -= ur[i] * uc[j];
        } else {
          a[kb][kb] -= ur[kb] * uc[kb];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) blk[(ty + 16 * i) * NB + tx + 16 * j] = a[i][j];
}

template <int MODE>
void run(double* S, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 20;
  hipLaunchKernelGGL(probe<MODE>, dim3(4), dim3(256), 0, 0, S, reps);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(4), dim3(256), 0, 0, S, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("mode %2d  %-58s %7.1f us per sweep  %6.0f ns per step\n", MODE, what, best * 1e3 / reps, best * 1e6 / reps / 128);
}

int main() {
  double* S;
  hipMalloc(&S, sizeof(double) * 4 * NB * NB);
  double* h = new double[4 * NB * NB];
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < NB; ++i)
      for (int j = 0; j < NB; ++j) h[(b * NB + i) * NB + j] = (i == j) ? 1e6 : 1e-3 / (1 + (i > j ? i - j : j - i));
  hipMemcpy(S, h, sizeof(double) * 4 * NB * NB, hipMemcpyHostToDevice);
  run<0>(S, "full step");
  run<1>(S, "no sqrt/divide");
  run<2>(S, "no barrier");
  run<4>(S, "no LDS broadcast");
  run<8>(S, "no trailing FMAs");
  run<9>(S, "no sqrt/divide, no FMAs");
  run<3>(S, "no sqrt/divide, no barrier");
  run<7>(S, "no sqrt/divide, no barrier, no LDS");
  run<15>(S, "only the shuffle + scale + loop");
  run<16>(S, "no owner branch (all threads do the pivot work)");
  return 0;
}
