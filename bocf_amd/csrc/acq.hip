// Acquisition kernels (closed-form maEI/maPI, Monte-Carlo uEI_noiseless/uPI), best-so-far
// reductions and the top-k selection.  All reductions use a fixed order (deterministic).
#include "bocf_internal.h"
#include <type_traits>
#include <cstdlib>
#include "../../include/bocf_hip.h"

#define PI_D 3.14159265358979323846

// U(theta, y): the utilities of the reference's experiment scripts (see include/bocf_hip.h)
__device__ __forceinline__ double utility_eval(int kind, const double* __restrict__ theta, const double* __restrict__ params,
                                               const double (&y)[BOCF_MAX_M], int m) {
  double acc = 0.0;
  if (kind == BOCF_UTIL_LINEAR) {
#pragma unroll
    for (int j = 0; j < BOCF_MAX_M; ++j)
      if (j < m) acc += theta[j] * y[j];
    return acc;
  }
  if (kind == BOCF_UTIL_NEG_SQ_DIST) {
#pragma unroll
    for (int j = 0; j < BOCF_MAX_M; ++j)
      if (j < m) {
        const double t = y[j] - theta[j];
        acc += t * t;
      }
    return -acc;
  }
  if (kind == BOCF_UTIL_NEG_SUM_EXP) {
#pragma unroll
    for (int j = 0; j < BOCF_MAX_M; ++j)
      if (j < m) acc += -exp(y[j]);
    return acc;
  }
  if (kind == BOCF_UTIL_NEG_EXP_COS) {
#pragma unroll
    for (int j = 0; j < BOCF_MAX_M; ++j)
      if (j < m) acc += params[j] * (exp(-y[j] / PI_D) * cos(PI_D * y[j]));
    return -acc;
  }
  // BOCF_UTIL_ROSENBROCK: val -= (a - y_j)^2 + 100 y_{j+h}^2, h = m/2 (test_5a.py:48-52)
  const int h = m >> 1;
  const double a = theta[0];
  double val = 0.0;
#pragma unroll
  for (int j = 0; j < BOCF_MAX_M / 2; ++j)
    if (j < h) {
      const double t = a - y[j];
      val -= t * t + 100.0 * (y[j + h] * y[j + h]);
    }
  return val;
}

// dU/dy: the analytic derivatives the experiment scripts pass as dfunc (test_1a.py:94-96,
// test_2a.py:64-65, test_3a.py:60-66, test_5a.py:54-59; linear: theta)
__device__ __forceinline__ void utility_grad(int kind, const double* __restrict__ theta, const double* __restrict__ params,
                                             const double (&y)[BOCF_MAX_M], int m, double (&g)[BOCF_MAX_M]) {
  const int h = m >> 1;
#pragma unroll
  for (int j = 0; j < BOCF_MAX_M; ++j) {
    double v = 0.0;
    if (j < m) {
      if (kind == BOCF_UTIL_LINEAR) v = theta[j];
      else if (kind == BOCF_UTIL_NEG_SQ_DIST) v = -2.0 * (y[j] - theta[j]);
      else if (kind == BOCF_UTIL_NEG_SUM_EXP) v = -exp(y[j]);
      else if (kind == BOCF_UTIL_NEG_EXP_COS) {
        const double e = exp(-y[j] / PI_D);
        v = params[j] * (PI_D * e * sin(PI_D * y[j]) + e * cos(PI_D * y[j]) / PI_D);
      } else {
        v = j < h ? 2.0 * (theta[0] - y[j]) : (j < 2 * h ? -200.0 * y[j] : 0.0);
      }
    }
    g[j] = v;
  }
}

// best_l = max_i U(theta_l, mu(X_i))  (uEI_noiseless.py:76; maEI.py:129-136 with the linear utility)
__global__ __launch_bounds__(256) void best_so_far_kernel(const double* __restrict__ mu_train, int N, int m, int util_kind,
                                                          const double* __restrict__ theta, int theta_dim,
                                                          const double* __restrict__ params, double* __restrict__ best) {
  const int l = blockIdx.x;
  const double* th = theta + (long)l * theta_dim;
  double mx = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 256) {
    double y[BOCF_MAX_M];
#pragma unroll
    for (int j = 0; j < BOCF_MAX_M; ++j) y[j] = j < m ? mu_train[(long)j * N + i] : 0.0;
    const double v = utility_eval(util_kind, th, params, y, m);
    mx = fmax(mx, v);
  }
  __shared__ double red[256];
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) best[l] = red[0];
}

void launch_best_so_far(const double* mu_train, int N, int m, int linear, int util_kind, const double* theta, int theta_dim, int L,
                        const double* util_params, double* best, hipStream_t s) {
  (void)linear;
  BOCF_LAUNCH(best_so_far_kernel, dim3((unsigned)L), dim3(256), 0, s, mu_train, N, m, util_kind, theta, theta_dim, util_params, best);
}

// Closed-form EI / PI of theta . f(x)  (maEI.py:81-98,147-163; maPI.py:78-94,138-158)
__global__ __launch_bounds__(256) void acq_linear_kernel(AcqArgs a) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= a.C) return;
  double mean[BOCF_MAX_M], var[BOCF_MAX_M];
#pragma unroll
  for (int j = 0; j < BOCF_MAX_M; ++j) {
    mean[j] = j < a.m ? a.mean[(long)j * a.ld + c] : 0.0;
    var[j] = j < a.m ? a.var[(long)j * a.ld + c] : 0.0;
  }
  double acq = 0.0;
  for (int l = 0; l < a.L; ++l) {
    const double* th = a.theta + (long)l * a.theta_dim;
    double mu = 0.0, s2 = 0.0;
#pragma unroll
    for (int j = 0; j < BOCF_MAX_M; ++j)
      if (j < a.m) {
        mu += th[j] * mean[j];
        s2 += (th[j] * th[j]) * var[j];
      }
    const double sigma = sqrt(s2);
    const double sfl = sigma < 1e-10 ? 1e-10 : sigma;          // floor only enters u (maEI.py:157-160)
    const double fmax_ = a.kind == BOCF_ACQ_EI ? a.best[l] : a.best[l] + 1e-6;
    const double u = (mu - fmax_) / sfl;
    const double Phi = 0.5 * erfc(-u / 1.41421356237309504880);
    double val;
    if (a.kind == BOCF_ACQ_EI) {
      const double phi = exp(-0.5 * u * u) / 2.50662827463100050242;   // sqrt(2 pi)
      val = sigma * (u * Phi + phi);
    } else {
      val = Phi;
    }
    acq += val * a.prob[l];
  }
  a.acq[c] = (a.accumulate ? a.acq[c] : 0.0) + acq * a.scale;
}

void launch_acq_linear(const AcqArgs& a, hipStream_t s) {
  if (a.C == 0) return;
  BOCF_LAUNCH(acq_linear_kernel, dim3((unsigned)((a.C + 255) / 256)), dim3(256), 0, s, a);
}

// Monte-Carlo EI / PI of a composite utility: one wave per candidate, lanes stride the S common
// random numbers (coalesced reads of the transposed normals), __shfl butterfly sum.
// (uEI_noiseless.py:63-83; uPI.py:66-86)
__global__ __launch_bounds__(256) void acq_mc_kernel(AcqArgs a) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= a.C) return;                         // wave-uniform
  double mu[BOCF_MAX_M], sg[BOCF_MAX_M];
#pragma unroll
  for (int j = 0; j < BOCF_MAX_M; ++j) {
    mu[j] = j < a.m ? a.mean[(long)j * a.ld + c] : 0.0;
    sg[j] = j < a.m ? sqrt(a.var[(long)j * a.ld + c]) : 0.0;
  }
  double acq = 0.0;
  for (int l = 0; l < a.L; ++l) {
    const double* th = a.theta + (long)l * a.theta_dim;
    const double best = a.kind == BOCF_ACQ_EI ? a.best[l] : a.best[l] + 1e-6;
    double part = 0.0;
    for (int s = lane; s < a.S; s += 64) {
      double y[BOCF_MAX_M];
#pragma unroll
      for (int j = 0; j < BOCF_MAX_M; ++j) y[j] = j < a.m ? mu[j] + sg[j] * a.Wt[(long)j * a.S + s] : 0.0;
      const double v = utility_eval(a.util_kind, th, a.util_params, y, a.m);
      if (a.kind == BOCF_ACQ_EI) part += fmax(v - best, 0.0);
      else part += (v - best) > 0.0 ? 1.0 : 0.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    acq += (part / (double)a.S) * a.prob[l];
  }
  if (lane == 0) a.acq[c] = (a.accumulate ? a.acq[c] : 0.0) + acq * a.scale;
}

// The same kernel with the output count as a template argument (m <= 8): the loops over the outputs -- building y, and inside the utility --
// unroll to exactly m terms instead of BOCF_MAX_M = 16 guarded ones (at m = 4 three quarters of the generic kernel's vector instructions
// were predicated-off iterations).  Same operations on the same values in the same order: the same bits.
template <int M>
__global__ __launch_bounds__(256) void acq_mc_m_kernel(AcqArgs a) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= a.C) return;                         // wave-uniform
  double mu[M], sg[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    mu[j] = a.mean[(long)j * a.ld + c];
    sg[j] = sqrt(a.var[(long)j * a.ld + c]);
  }
  double acq = 0.0;
  for (int l = 0; l < a.L; ++l) {
    const double* th = a.theta + (long)l * a.theta_dim;
    const double best = a.kind == BOCF_ACQ_EI ? a.best[l] : a.best[l] + 1e-6;
    double part = 0.0;
    for (int s = lane; s < a.S; s += 64) {
      double y[BOCF_MAX_M];
#pragma unroll
      for (int j = 0; j < BOCF_MAX_M; ++j) y[j] = 0.0;
#pragma unroll
      for (int j = 0; j < M; ++j) y[j] = mu[j] + sg[j] * a.Wt[(long)j * a.S + s];
      const double v = utility_eval(a.util_kind, th, a.util_params, y, M);
      if (a.kind == BOCF_ACQ_EI) part += fmax(v - best, 0.0);
      else part += (v - best) > 0.0 ? 1.0 : 0.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    acq += (part / (double)a.S) * a.prob[l];
  }
  if (lane == 0) a.acq[c] = (a.accumulate ? a.acq[c] : 0.0) + acq * a.scale;
}

void launch_acq_mc(const AcqArgs& a, hipStream_t s) {
  if (a.C == 0) return;
  const dim3 grid((unsigned)((a.C + 3) / 4));
  if (a.m >= 1 && a.m <= 8) {
#define LM(M) case M: BOCF_LAUNCH((acq_mc_m_kernel<M>), grid, dim3(256), 0, s, a); break;
    switch (a.m) { LM(1) LM(2) LM(3) LM(4) LM(5) LM(6) LM(7) LM(8) default: break; }
#undef LM
    return;
  }
  BOCF_LAUNCH(acq_mc_kernel, grid, dim3(256), 0, s, a);
}

// Closed-form EI / PI with input gradients (maEI.py:101-126, maPI.py:96-121): thread per candidate.
__global__ __launch_bounds__(256) void acq_linear_grad_kernel(AcqArgs a) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= a.C) return;
  double mean[BOCF_MAX_M], var[BOCF_MAX_M];
#pragma unroll
  for (int j = 0; j < BOCF_MAX_M; ++j) {
    mean[j] = j < a.m ? a.mean[(long)j * a.ld + c] : 0.0;
    var[j] = j < a.m ? a.var[(long)j * a.ld + c] : 0.0;
  }
  double acq = 0.0;
  if (!a.accumulate)
    for (int q = 0; q < a.d; ++q) a.dacq[(long)c * a.d + q] = 0.0;
  for (int l = 0; l < a.L; ++l) {
    const double* th = a.theta + (long)l * a.theta_dim;
    double mu = 0.0, s2 = 0.0;
#pragma unroll
    for (int j = 0; j < BOCF_MAX_M; ++j)
      if (j < a.m) {
        mu += th[j] * mean[j];
        s2 += (th[j] * th[j]) * var[j];
      }
    const double sigma = sqrt(s2);
    double Phi, phi, u, val;
    if (a.kind == BOCF_ACQ_EI) {            // norm.cdf / norm.pdf on the unfloored sigma (maEI.py:117-119)
      u = (mu - a.best[l]) / sigma;
      Phi = 0.5 * erfc(-u / 1.41421356237309504880);
      phi = exp(-0.5 * u * u) / 2.50662827463100050242;
      val = (mu - a.best[l]) * Phi + sigma * phi;
    } else {                                // _get_quantiles with floor and jitter (maPI.py:112-116)
      const double sfl = sigma < 1e-10 ? 1e-10 : sigma;
      u = (mu - (a.best[l] + 1e-6)) / sfl;
      Phi = 0.5 * erfc(-u / 1.41421356237309504880);
      phi = exp(-0.5 * u * u) / 2.50662827463100050242;
      val = Phi;
    }
    acq += val * a.prob[l];
    for (int q = 0; q < a.d; ++q) {
      double dmu = 0.0, dv = 0.0;
#pragma unroll
      for (int j = 0; j < BOCF_MAX_M; ++j)
        if (j < a.m) {
          dmu += th[j] * a.dmean[((long)j * a.ldg + c) * a.d + q];
          dv += (th[j] * th[j]) * a.dvar[((long)j * a.ldg + c) * a.d + q];
        }
      const double dsig = 0.5 * dv / sigma;
      const double g = a.kind == BOCF_ACQ_EI ? dmu * Phi + phi * dsig : (phi / sigma) * (dmu - u * dsig);
      a.dacq[(long)c * a.d + q] += (g * a.prob[l]) * a.scale;
    }
  }
  a.acq[c] = (a.accumulate ? a.acq[c] : 0.0) + acq * a.scale;
}

void launch_acq_linear_grad(const AcqArgs& a, hipStream_t s) {
  if (a.C == 0) return;
  BOCF_LAUNCH(acq_linear_grad_kernel, dim3((unsigned)((a.C + 255) / 256)), dim3(256), 0, s, a);
}

// Monte-Carlo EI with input gradients (uEI_noiseless.py:138-170): wave per candidate.  For every improving
// sample (U > best, strict) accumulate A_j += dU/dy_j and B_j += dU/dy_j * 0.5 W_sj / sigma_j; then
// d acq/dx_q = (1/S) sum_j A_j dmu_j/dx_q + B_j dvar_j/dx_q.
// (MC = the output count as a template argument, 1 ... 8, or 0 = read it from the arguments: with a constant every loop over the outputs
//  below and inside the utility unrolls to exactly m terms -- see acq_mc_m_kernel)
template <int MC>
__global__ __launch_bounds__(256) void acq_mc_grad_kernel(AcqArgs a) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= a.C) return;
  const int m_ = MC > 0 ? MC : a.m;
  constexpr int MM = MC > 0 ? MC : BOCF_MAX_M;           // loop bound of the per-output loops
  double mu[BOCF_MAX_M], sg[BOCF_MAX_M];
#pragma unroll
  for (int j = 0; j < BOCF_MAX_M; ++j) {
    mu[j] = j < m_ ? a.mean[(long)j * a.ld + c] : 0.0;
    sg[j] = j < m_ ? sqrt(a.var[(long)j * a.ld + c]) : 1.0;
  }
  double acq = 0.0;
  double dq = 0.0;                                    // lane q < d accumulates d acq / dx_q
  for (int l = 0; l < a.L; ++l) {
    const double* th = a.theta + (long)l * a.theta_dim;
    const double best = a.best[l];
    double part = 0.0;
    double A[BOCF_MAX_M], Bc[BOCF_MAX_M];
#pragma unroll
    for (int j = 0; j < BOCF_MAX_M; ++j) { A[j] = 0.0; Bc[j] = 0.0; }
    for (int s = lane; s < a.S; s += 64) {
      double y[BOCF_MAX_M], w[BOCF_MAX_M], g[BOCF_MAX_M];
#pragma unroll
      for (int j = 0; j < BOCF_MAX_M; ++j) {
        w[j] = j < m_ ? a.Wt[(long)j * a.S + s] : 0.0;
        y[j] = mu[j] + sg[j] * w[j];
      }
      const double v = utility_eval(a.util_kind, th, a.util_params, y, m_);
      part += fmax(v - best, 0.0);
      if (v > best) {
        utility_grad(a.util_kind, th, a.util_params, y, m_, g);
#pragma unroll
        for (int j = 0; j < MM; ++j) {
          A[j] += g[j];
          Bc[j] += g[j] * (0.5 * w[j] / sg[j]);
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
#pragma unroll
    for (int j = 0; j < MM; ++j) {
      if (j < m_) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          A[j] += __shfl_xor(A[j], o, 64);
          Bc[j] += __shfl_xor(Bc[j], o, 64);
        }
      }
    }
    const double wgt = a.prob[l] / (double)a.S;
    acq += part * wgt;
    if (lane < a.d) {
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < MM; ++j)
        if (j < m_) t += A[j] * a.dmean[((long)j * a.ldg + c) * a.d + lane] + Bc[j] * a.dvar[((long)j * a.ldg + c) * a.d + lane];
      dq += t * wgt;
    }
  }
  if (lane == 0) a.acq[c] = (a.accumulate ? a.acq[c] : 0.0) + acq * a.scale;
  if (lane < a.d) a.dacq[(long)c * a.d + lane] = (a.accumulate ? a.dacq[(long)c * a.d + lane] : 0.0) + dq * a.scale;
}

void launch_acq_mc_grad(const AcqArgs& a, hipStream_t s) {
  if (a.C == 0) return;
  const dim3 grid((unsigned)((a.C + 3) / 4));
  if (a.m >= 1 && a.m <= 8) {
#define LM(M) case M: BOCF_LAUNCH((acq_mc_grad_kernel<M>), grid, dim3(256), 0, s, a); break;
    switch (a.m) { LM(1) LM(2) LM(3) LM(4) LM(5) LM(6) LM(7) LM(8) default: break; }
#undef LM
    return;
  }
  BOCF_LAUNCH((acq_mc_grad_kernel<0>), grid, dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
// top-k: strict total order (value descending, index ascending); round t takes the best element
// that is strictly worse than round t-1's winner, so no "taken" flags are needed.
// np.argsort(-acq)[:k] (anchor_points_generator.py:61) with ties to the lowest index.
struct VI { double v; long long i; };
__device__ __forceinline__ bool better(double av, long long ai, double bv, long long bi) {
  return av > bv || (av == bv && ai < bi);
}

// Streaming form for blocks too large for registers (more than 16 elements per thread: beyond 262144 candidates per context, or more
// than 4096 gathered winners): every round re-reads the block; one barrier per round.
template <int NT = 256>
__global__ __launch_bounds__(NT) void topk_stream_kernel(const double* __restrict__ vals, const long long* __restrict__ idxs, long long n,
                                                          long long per_block, int k, long long* __restrict__ out_idx,
                                                          double* __restrict__ out_val) {
  const long long lo = (long long)blockIdx.x * per_block;
  long long hi = lo + per_block;
  if (hi > n) hi = n;
  constexpr int NW = NT / 64;
  __shared__ double rv[2][NW];
  __shared__ long long ri[2][NW];
  const long long NONE = 0x7fffffffffffffffLL;
  double pv = INFINITY;
  long long pi = -1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int t = 0; t < k; ++t) {
    double bv = -INFINITY;
    long long bi = NONE;
    for (long long e = lo + threadIdx.x; e < hi; e += NT) {
      double v = vals[e];
      if (!(v == v)) v = -INFINITY;                        // NaN sorts last
      const long long id = idxs ? idxs[e] : e;
      if (id < 0) continue;                                // empty slot of a previous stage
      const bool worse_than_prev = v < pv || (v == pv && id > pi);
      if (worse_than_prev && better(v, id, bv, bi)) { bv = v; bi = id; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(bv, o, 64);
      const long long oi = __shfl_xor(bi, o, 64);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { rv[t & 1][w] = bv; ri[t & 1][w] = bi; }
    __syncthreads();
    double fv = rv[t & 1][0];
    long long fi = ri[t & 1][0];
#pragma unroll
    for (int q = 1; q < NW; ++q) if (better(rv[t & 1][q], ri[t & 1][q], fv, fi)) { fv = rv[t & 1][q]; fi = ri[t & 1][q]; }
    const bool found = fi != NONE;
    if (threadIdx.x == 0) {
      out_idx[(long long)blockIdx.x * k + t] = found ? fi : -1;
      out_val[(long long)blockIdx.x * k + t] = found ? fv : -INFINITY;
    }
    pv = fv; pi = fi;
    if (!found) {                                          // exhausted (every thread sees it): fill the rest
      if (threadIdx.x == 0)
        for (int q = t + 1; q < k; ++q) { out_idx[(long long)blockIdx.x * k + q] = -1; out_val[(long long)blockIdx.x * k + q] = -INFINITY; }
      break;
    }
  }
}

// max of a double over the 64 lanes of a wave through DPP row operations (no LDS round trip: a ds_bpermute butterfly costs ~1.2 us per
// (value, index) reduction, which is what the k dependent rounds of a selection are made of).  The classic gfx9 sequence: two quad
// permutes, half-row mirror, row mirror (every lane of a 16-lane row then holds the row's max), row broadcasts 15 and 31; lane 63 holds
// the wave's max and is read back as a scalar.  Inputs must not be NaN.
__device__ __forceinline__ double dpp_step_max(double v, const int ctrl, const int row_mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  int olo, ohi;
  switch (ctrl) {          // (the control word must be a literal)
    case 0xB1: olo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;
    case 0x4E: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;
    case 0x141: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break;
    case 0x140: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break;
    case 0x142: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0xA, 0xF, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0xA, 0xF, false); break;
    default: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x143, 0xC, 0xF, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x143, 0xC, 0xF, false); break;
  }
  (void)row_mask;
  const double o = __hiloint2double(ohi, olo);
  return o > v ? o : v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
  v = dpp_step_max(v, 0xB1, 0xF);
  v = dpp_step_max(v, 0x4E, 0xF);
  v = dpp_step_max(v, 0x141, 0xF);
  v = dpp_step_max(v, 0x140, 0xF);
  v = dpp_step_max(v, 0x142, 0xA);
  v = dpp_step_max(v, 0x143, 0xC);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// Selection with the k dependent rounds INSIDE a wave (round 4).  What the rounds cost is the scan of a lane's register slots (k rounds x E
// slots x ~14 vector instructions, on ONE compute unit when one workgroup holds the whole vector: 60 us for the 8192 candidates of BASELINE
// configs[1], a tenth of that step, with or without a barrier per round) -- so the vector is spread over MANY small workgroups with few
// slots per lane: 256 threads x E slots each (E = 1 at 8192 candidates: 32 workgroups), every wave selects the k best of its 64 E elements
// with shuffles only (no LDS, no barrier), the 4 x k wave winners meet in LDS once and wave 0 selects the block's k; a second launch of the
// same kernel merges the blocks' winners.  ~10 us in all.  Same strict total order (value descending, index ascending), same outputs.
template <int E>
__global__ __launch_bounds__(256) void topk_wave_kernel(const double* __restrict__ vals, const long long* __restrict__ idxs, long long n,
                                                        long long per_block, int k, long long* __restrict__ out_idx,
                                                        double* __restrict__ out_val) {
  constexpr int NW = 4;
  __shared__ double wv[NW * 64];                           // k <= 64 winners per wave
  __shared__ long long wi[NW * 64];
  const long long lo = (long long)blockIdx.x * per_block;
  long long hi = lo + per_block;
  if (hi > n) hi = n;
  const long long NONE = 0x7fffffffffffffffLL;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double cv[E];
  long long ci[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const long long pos = lo + threadIdx.x + 256 * e;
    double v = pos < hi ? vals[pos] : -INFINITY;
    if (!(v == v)) v = -INFINITY;                          // NaN sorts last
    cv[e] = v;
    ci[e] = pos < hi ? (idxs ? idxs[pos] : pos) : -1;      // id < 0: empty slot (of a previous stage, or past the end)
  }
  // k rounds over the register slots of every lane of this wave: the best element strictly worse than the previous winner
  auto select = [&](auto& xv, auto& xi, auto nslots, double* ov, long long* oi) {
    double pv = INFINITY;
    long long pi = -1;
    for (int t = 0; t < k; ++t) {
      double bv = -INFINITY;
      long long bi = NONE;
#pragma unroll
      for (int e = 0; e < decltype(nslots)::value; ++e) {
        const bool worse_than_prev = xv[e] < pv || (xv[e] == pv && xi[e] > pi);
        if (xi[e] >= 0 && worse_than_prev && better(xv[e], xi[e], bv, bi)) { bv = xv[e]; bi = xi[e]; }
      }
      // the wave's winner: max value (DPP), then -- among the lanes that hold it -- the lowest index, again as a DPP max (of the negated
      // index: candidate indices are exact in a double, below 2^53); acquisition vectors are full of exact ties (zeros)
      const double mx = wave_max_f64(bv);
      const double neg_i = wave_max_f64(bi != NONE && bv == mx ? -(double)bi : -INFINITY);
      const bool found = neg_i > -INFINITY;
      bv = mx;
      bi = found ? (long long)(-neg_i) : NONE;
      if (lane == 0) {
        ov[t] = found ? bv : -INFINITY;
        oi[t] = found ? bi : -1;
      }
      pv = bv; pi = bi;
      if (!found) {                                        // exhausted (uniform across the wave): fill the rest
        if (lane == 0)
          for (int q = t + 1; q < k; ++q) { ov[q] = -INFINITY; oi[q] = -1; }
        break;
      }
    }
  };
  select(cv, ci, std::integral_constant<int, E>(), wv + w * 64, wi + w * 64);
  __syncthreads();
  if (w == 0) {
    // the 4 x k wave winners (k <= 64: at most 4 slots per lane): slot e of lane l = winner (e * 64 + l) % k of wave (e * 64 + l) / k
    double mv[4];
    long long mi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = e * 64 + lane;
      const bool in = q < NW * k;
      mv[e] = in ? wv[(q / k) * 64 + q % k] : -INFINITY;
      mi[e] = in ? wi[(q / k) * 64 + q % k] : -1;
    }
    select(mv, mi, std::integral_constant<int, 4>(), out_val + (long long)blockIdx.x * k, out_idx + (long long)blockIdx.x * k);
  }
}

// blocks of the first stage: as many as keep the slots per lane low, at most 256 (so that the second stage holds their <= 256 x 64 winners
// in <= 16 slots per lane... for k = 16: 4096 winners)
int topk_num_blocks(int C) {
  int nb = (C + 255) / 256;                                // one slot per lane
  if (nb > 64) nb = 64;
  if (nb < 1) nb = 1;
  return nb;
}

template <typename... A>
static void launch_topk_wave(int slots, dim3 grid, hipStream_t s, A... args) {
  if (slots <= 1) BOCF_LAUNCH(topk_wave_kernel<1>, grid, dim3(256), 0, s, args...);
  else if (slots <= 2) BOCF_LAUNCH(topk_wave_kernel<2>, grid, dim3(256), 0, s, args...);
  else if (slots <= 4) BOCF_LAUNCH(topk_wave_kernel<4>, grid, dim3(256), 0, s, args...);
  else if (slots <= 8) BOCF_LAUNCH(topk_wave_kernel<8>, grid, dim3(256), 0, s, args...);
  else BOCF_LAUNCH(topk_wave_kernel<16>, grid, dim3(256), 0, s, args...);
}

void launch_topk(const double* acq, int C, int k, long long* blk_idx, double* blk_val, long long* out_idx, double* out_val, hipStream_t s) {
  const int nb = topk_num_blocks(C);
  const long long per = ((long long)C + nb - 1) / nb;
  if (nb == 1) {
    launch_topk_wave((int)((per + 255) / 256), dim3(1), s, acq, (const long long*)nullptr, (long long)C, per, k, out_idx, out_val);
    return;
  }
  if (per <= 4096) launch_topk_wave((int)((per + 255) / 256), dim3((unsigned)nb), s, acq, (const long long*)nullptr, (long long)C, per, k, blk_idx, blk_val);
  else BOCF_LAUNCH(topk_stream_kernel<256>, dim3((unsigned)nb), dim3(256), 0, s, acq, (const long long*)nullptr, (long long)C, per, k, blk_idx, blk_val);
  // nb * k <= 64 * 64 = 4096 winners
  launch_topk_wave((int)(((long long)nb * k + 255) / 256), dim3(1), s, (const double*)blk_val, (const long long*)blk_idx, (long long)nb * k, (long long)nb * k, k,
                   out_idx, out_val);
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU selection: rank r's k local winners go into slots [r k, (r+1) k) of a 2 G k buffer of doubles --
// values in the first half, GLOBAL candidate indices (lo + local index, exact below 2^53) in the second, -inf everywhere
// else -- so that ONE all-reduce(MAX) assembles every rank's winners on every rank (RCCL has no MAXLOC).
__global__ void pack_topk_kernel(const long long* __restrict__ idx, const double* __restrict__ val, int k, long long lo, int world, int rank,
                                 double* __restrict__ pack) {
  const int n = world * k;
  for (int t = threadIdx.x; t < 2 * n; t += blockDim.x) {
    const int slot = t < n ? t : t - n;
    double v = -INFINITY;
    const int q = slot - rank * k;
    if (idx && q >= 0 && q < k && idx[q] >= 0) v = t < n ? val[q] : (double)(idx[q] + lo);
    pack[t] = v;
  }
}

void launch_pack_topk(const long long* idx, const double* val, int k, long long lo, int world, int rank, double* pack, hipStream_t s) {
  BOCF_LAUNCH(pack_topk_kernel, dim3(1), dim3(256), 0, s, idx, val, k, lo, world, rank, pack);
}

__global__ void unpack_topk_kernel(const double* __restrict__ pack, int n, long long* __restrict__ gidx, double* __restrict__ gval) {
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const double id = pack[n + t];
    const bool ok = id >= 0.0 && id < 9007199254740992.0;    // finite (an empty slot holds -inf)
    gidx[t] = ok ? (long long)id : -1;
    gval[t] = ok ? pack[t] : -INFINITY;
  }
}

// the G k gathered pairs -> the global k best (value descending, index ascending: np.argsort(-acq)[:k] on the whole batch)
void launch_merge_packed(const double* pack, int k, int world, long long* gidx, double* gval, long long* out_idx, double* out_val, hipStream_t s) {
  const int n = world * k;
  BOCF_LAUNCH(unpack_topk_kernel, dim3(1), dim3(256), 0, s, pack, n, gidx, gval);
  if (n <= 4096) launch_topk_wave((n + 255) / 256, dim3(1), s, (const double*)gval, (const long long*)gidx, (long long)n, (long long)n, k, out_idx, out_val);
  else BOCF_LAUNCH(topk_stream_kernel<256>, dim3(1), dim3(256), 0, s, (const double*)gval, (const long long*)gidx, (long long)n, (long long)n, k, out_idx, out_val);
}
