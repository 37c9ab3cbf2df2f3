// Where do the ~370 ns per elimination step of the 128x128 diagonal-block kernel go?  The unblocked register-cyclic
// Cholesky sweep (potrf_diag_kernel's phase 1) with parts switched off (results are then wrong -- timing only):
//   bit 0: no IEEE sqrt / divide (a multiply instead)   bit 1: no workgroup barrier   bit 2: no LDS broadcast (registers)
//   bit 3: no trailing FMAs                               bit 4: no owner branch (every thread does the pivot work)
//   hipcc --offload-arch=gfx950 -O3 tools/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <type_traits>
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


// Dataflow variant: no workgroup barrier inside the sweep.  Finished rows live in LDS (the U image the inverse phase needs
// anyway) with one ready flag each; the wave that owns row k+1 applies update k to that row first, produces and publishes
// row k+1, and only then finishes update k on its other registers; the other waves poll the flag.  Bounded spins: a bug
// can produce wrong numbers but not a hang.
__device__ __forceinline__ void wait_flag(volatile int* f) {
  int spins = 0;
  while (*f == 0 && ++spins < (1 << 22)) {}
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#define PRODUCE(KB, KK)                                                                  \
  do {                                                                                   \
    const int kk_ = (KK), k_ = (KB) * 16 + kk_;                                          \
    double piv_ = __shfl(a[KB][KB], (kk_ & 3) * 16 + kk_, 64);                           \
    if (!(piv_ > 0.0)) piv_ = 1.0;                                                       \
    const double ukk_ = sqrt(piv_);                                                      \
    const double inv_ = 1.0 / ukk_;                                                      \
    if (ty == kk_) {                                                                     \
      _Pragma("unroll") for (int j_ = (KB); j_ < 8; ++j_) {                              \
        const int c_ = tx + 16 * j_;                                                     \
        const double v_ = a[KB][j_] * inv_;                                              \
        if (c_ > k_) a[KB][j_] = v_;                                                     \
        else if (c_ == k_) a[KB][j_] = ukk_;                                             \
        Ul[k_ * 129 + c_] = (c_ > k_) ? v_ : 0.0;                                        \
      }                                                                                  \
    }                                                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                               \
    if (lane == 0) *(volatile int*)&flags[k_] = 1;                                       \
  } while (0)

template <int kb>
__device__ __forceinline__ void sweep_block(double (&a)[8][8], double* Ul, int* flags, int ty, int tx, int wave, int lane) {
#pragma unroll 1
  for (int kk = 0; kk < 15; ++kk) {                        // rows whose successor is in the same 16-row block
    const int k = kb * 16 + kk;
    if (wave != (kk >> 2)) wait_flag(&flags[k]);             // the owner wave wrote it itself (program order + fence)
    const double* rb = Ul + k * 129;
    double ur[8], uc[8];
#pragma unroll
    for (int i = kb; i < 8; ++i) {
      ur[i] = rb[ty + 16 * i];
      uc[i] = rb[tx + 16 * i];
    }
#pragma unroll
    for (int j = kb; j < 8; ++j) a[kb][j] -= ur[kb] * uc[j];            // the block's own rows first: row k+1 is among them
    if (wave == ((kk + 1) >> 2)) PRODUCE(kb, kk + 1);
#pragma unroll
    for (int i = kb + 1; i < 8; ++i)
#pragma unroll
      for (int j = kb; j < 8; ++j) a[i][j] -= ur[i] * uc[j];
  }
  if constexpr (kb < 7) {                                    // last row of the block: its successor opens the next block
    const int k = kb * 16 + 15;
    if (wave != 3) wait_flag(&flags[k]);
    const double* rb = Ul + k * 129;
    double ur[8], uc[8];
#pragma unroll
    for (int i = kb + 1; i < 8; ++i) {
      ur[i] = rb[ty + 16 * i];
      uc[i] = rb[tx + 16 * i];
    }
#pragma unroll
    for (int j = kb + 1; j < 8; ++j) a[kb + 1][j] -= ur[kb + 1] * uc[j];
    if (wave == 0) PRODUCE(kb + 1, 0);
#pragma unroll
    for (int i = kb + 2; i < 8; ++i)
#pragma unroll
      for (int j = kb + 1; j < 8; ++j) a[i][j] -= ur[i] * uc[j];
  }
}

__global__ __launch_bounds__(256, 1) void probe_dataflow(double* S, int reps) {
  __shared__ double Ul[NB * 129];
  __shared__ int flags[NB];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15, wave = ty >> 2, lane = tid & 63;
  double* blk = S + (long)blockIdx.x * NB * NB;
  double a[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a[i][j] = blk[(ty + 16 * i) * NB + tx + 16 * j];
  for (int rep = 0; rep < reps; ++rep) {
    __syncthreads();
    if (tid < NB) flags[tid] = 0;
    __syncthreads();
    // produce row `kb*16 + kk` from register row kb (all of this wave's prior updates to it are done)
    if (wave == 0) PRODUCE(0, 0);
    sweep_block<0>(a, Ul, flags, ty, tx, wave, lane);
    sweep_block<1>(a, Ul, flags, ty, tx, wave, lane);
    sweep_block<2>(a, Ul, flags, ty, tx, wave, lane);
    sweep_block<3>(a, Ul, flags, ty, tx, wave, lane);
    sweep_block<4>(a, Ul, flags, ty, tx, wave, lane);
    sweep_block<5>(a, Ul, flags, ty, tx, wave, lane);
    sweep_block<6>(a, Ul, flags, ty, tx, wave, lane);
    sweep_block<7>(a, Ul, flags, ty, tx, wave, lane);
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
  // dataflow variant: same numbers as the barrier sweep on the upper triangle?
  double* r0 = new double[4 * NB * NB];
  double* r1 = new double[4 * NB * NB];
  hipMemcpy(S, h, sizeof(double) * 4 * NB * NB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe<0>, dim3(4), dim3(256), 0, 0, S, 1);
  hipMemcpy(r0, S, sizeof(double) * 4 * NB * NB, hipMemcpyDeviceToHost);
  hipMemcpy(S, h, sizeof(double) * 4 * NB * NB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe_dataflow, dim3(4), dim3(256), 0, 0, S, 1);
  hipMemcpy(r1, S, sizeof(double) * 4 * NB * NB, hipMemcpyDeviceToHost);
  double maxd = 0.0;
  long bad = 0;
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < NB; ++i)
      for (int j = i; j < NB; ++j) {
        const double d = fabs(r0[(b * NB + i) * NB + j] - r1[(b * NB + i) * NB + j]);
        if (d > maxd) maxd = d;
        if (d != 0.0) ++bad;
      }
  printf("dataflow vs barrier sweep, upper triangle: max |diff| %.3e, %ld differing elements\n", maxd, bad);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe_dataflow, dim3(4), dim3(256), 0, 0, S, 20);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("dataflow sweep (flags instead of barriers)                          %7.1f us per sweep  %6.0f ns per step\n", best * 1e3 / 20, best * 1e6 / 20 / 128);
  return 0;
}
