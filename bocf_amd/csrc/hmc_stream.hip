// Stream-resident HMC over the hyper-parameters of models the fused chain (hmc128_kernel, fit.hip) does not serve (N > 128 or d > 16):
// GPy/inference/mcmc/hmc.py:30-69 with M = I as GPModel.updateModel runs it (gpmodel.py:117-118).  The inference of a leapfrog step is
// the ordinary launch sequence of bocf_fit + bocf_lml_gradients (kernel matrix, factorization, inverse, alpha, log-marginal, Ky^-1,
// hyper-gradient sums); what used to be a host round trip per step -- the O(P) arithmetic of hyper.py's lockstep loop -- is the kernel
// below, launched in front of and behind each inference on the same stream:
//   PRE   (first step of a draw: momenta, H_old, chain record;) p += -eps/2 g;  x += eps p;  theta = Logexp.f(x);  hyper-parameters -> device
//   POST  objective -(log-marginal + log-prior) and its gradient w.r.t. the optimizer array (paramz Model._objective_grads; Gamma priors
//         priors.py:264-330);  p += -eps/2 g;  (last step of a draw: Hamiltonian, Metropolis test hmc.py:51-58, accept / restore)
// with the arithmetic of hmc128_kernel operation for operation (parameter k on lane k, sums on lane 0 in hyper.py's order, no contraction).
// What the device cannot do inside a stream is jitchol's ladder (linalg.py:52-71: rebuild and refactor with jitter): a non-positive
// pivot, parameters that leave the positive domain, or a schedule time-out set *abort_draw = the draw they happened in; every later
// PRE / POST is then a no-op, FINISH puts every output back to where that draw started, and the HOST runs that one draw with the lockstep
// loop (ladder included) before the stream takes over again.
#include "fit_device.h"

#define HS_MAXP (2 + BOCF_MAX_D)
#define HS_STRIDE (7 * HS_MAXP + 8)      // th | x | x_old | pm | tg | tg_old | th_start | obj, obj_old, H_old, diverged

__device__ __forceinline__ void hs_write_hyp(const HmcStreamArgs& a, int jo, const double* th) {
  KernHyp* h = a.hyp + jo;
  h->variance = th[0];
  for (int q = 0; q < BOCF_MAX_D; ++q) h->ls[q] = q < a.d ? th[1 + (a.nls == 1 ? 0 : q)] : 1.0;
  h->noise = th[a.P - 1];
  h->jitter = -a.diag_shift;
  a.info[jo] = 0;
}

// what the host path launches in front of a factorization, for this output: the inputs divided by the lengthscales (scale_inputs_kernel,
// stationary.py:161-164) and the zeroed dependency counters of the team schedule (hipMemsetAsync there).  th: the parameters just written.
__device__ __forceinline__ void hs_stage(const HmcStreamArgs& a, int jo, const double* th) {
  const int d = a.d;
  double* Xs = a.Xs + (long)jo * a.strideXs;
  for (int idx = threadIdx.x; idx < a.N * d; idx += 256) {
    const int q = idx % d;
    Xs[idx] = a.X[idx] / th[1 + (a.nls == 1 ? 0 : q)];
  }
  if (a.flags) {
    int* F = a.flags + (long)jo * a.flag_words;
    for (int k = threadIdx.x; k < a.flag_words; k += 256) F[k] = 0;
    if (jo == 0 && threadIdx.x < 4) a.flags[(long)a.m * a.flag_words + threadIdx.x] = 0;
  }
}

__global__ __launch_bounds__(256) void hmc_stream_kernel(HmcStreamArgs a, int mode, int i, int it) {
  __shared__ double lpt[HS_MAXP], ljt[HS_MAXP], tgt[HS_MAXP];
  __shared__ int kfree[HS_MAXP];
  __shared__ double r1[256], r2[256];
  __shared__ double res_s[2 + BOCF_MAX_D], lml_s;
  const int jo = blockIdx.x, tid = threadIdx.x;
  const int P = a.P, d = a.d, nls = a.nls;
  const int* fx = a.fixed + (long)jo * P;
  double* st = a.state + (long)jo * HS_STRIDE;
  double* th = st;
  double* x = st + HS_MAXP;
  double* x_old = st + 2 * HS_MAXP;
  double* pm = st + 3 * HS_MAXP;
  double* tg = st + 4 * HS_MAXP;
  double* tg_old = st + 5 * HS_MAXP;
  double* th_start = st + 6 * HS_MAXP;
  double* sc = st + 7 * HS_MAXP;                         // obj, obj_old, H_old, diverged, accepted / diverged counts at the start of the draw
  const double half_log_2pi = 0.91893853320467274178;
  int Pf = 0;
  for (int k = 0; k < P; ++k) Pf += fx[k] ? 0 : 1;
  if (tid == 0) {
    int kf = 0;
    for (int k = 0; k < P; ++k)
      if (!fx[k]) kfree[kf++] = k;
  }
  __syncthreads();
  const int aborted = __hip_atomic_load(a.abort_draw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  auto set_free_from_x = [&](const double* xv) {           // o.optimizer_array = x: param_array[free] = Logexp.f(x)
    int kf = 0;
    for (int k = 0; k < P; ++k)
      if (!fx[k]) th[k] = hmc_logexp_f(xv[kf++]);
  };
  auto give_up = [&](int draw) {                           // this draw needs the host
    int expected = -1;
    __hip_atomic_compare_exchange_strong(a.abort_draw, &expected, draw, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (mode == HS_INIT) {
    if (tid == 0) {
      for (int k = 0; k < P; ++k) {
        th[k] = a.theta[(long)jo * P + k];
        th_start[k] = th[k];
      }
      sc[3] = 0.0;
      hs_write_hyp(a, jo, th);
    }
    __syncthreads();
    hs_stage(a, jo, th);
    return;
  }
  if (mode == HS_FINISH) {
    if (tid == 0) {
      if (aborted >= 0) {                                  // every output back to the start of the aborted draw, exactly
        for (int k = 0; k < P; ++k) th[k] = th_start[k];
        a.accepted[jo] = (int)sc[4];
        a.diverged[jo] = (int)sc[5];
      }
      for (int k = 0; k < P; ++k) a.theta[(long)jo * P + k] = th[k];
    }
    return;
  }
  if (aborted >= 0) return;
  if (mode == HS_PRE) {
    if (it == 0 && tid == 0) {
#pragma clang fp contract(off)
      const double* mi = a.mom + ((long)jo * a.ns + i) * P;
      double pp = 0.0;
      for (int k = 0; k < Pf; ++k) {
        pm[k] = mi[k];
        pp += pm[k] * pm[k];
      }
      sc[2] = sc[0] + Pf * half_log_2pi + pp / 2.0;        // H_old
      int kf = 0;
      for (int k = 0; k < P; ++k) {
        th_start[k] = th[k];
        if (!fx[k]) {
          x_old[kf] = hmc_logexp_finv(th[k]);
          x[kf] = x_old[kf];
          a.chains[((long)jo * a.ns + i) * P + kf] = th[k];
          ++kf;
        }
      }
      sc[1] = sc[0];
      for (int k = 0; k < Pf; ++k) tg_old[k] = tg[k];
      sc[3] = 0.0;
      sc[4] = (double)a.accepted[jo];                       // (an aborted draw is the host's: its counts are taken back)
      sc[5] = (double)a.diverged[jo];
    }
    __syncthreads();
    if (tid < Pf) {                                        // (free parameter kf on lane kf: independent updates)
#pragma clang fp contract(off)
      const double h = -a.eps / 2.0;
      const int k = tid;
      pm[k] += h * tg[k];
      x[k] += a.eps * pm[k];
      th[kfree[k]] = hmc_logexp_f(x[k]);
    }
    __syncthreads();
    if (tid == 0) {
      bool ok = true;
      for (int k = 0; k < P; ++k) ok = ok && isfinite(th[k]) && (k == P - 1 ? th[k] >= 0.0 : th[k] > 0.0);
      if (!ok) {
        give_up(i);
        for (int k = 0; k < P; ++k) th[k] = th_start[k];   // (keeps the launches in between on valid hyper-parameters)
      }
      hs_write_hyp(a, jo, th);
    }
    __syncthreads();
    hs_stage(a, jo, th);
    return;
  }
  // HS_EVAL0 (the objective at the chain's start) / HS_POST
  if (a.info[jo] != 0 || (a.sched_err && __hip_atomic_load(a.sched_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
    if (tid == 0) give_up(mode == HS_EVAL0 ? 0 : i);
    return;
  }
  {
    // log-marginal (lml_kernel) and the reduction of the hyper-gradient partials (hypgrad_reduce_kernel), same arithmetic and order
    double ld = 0.0, dt = 0.0;
    for (int r = tid; r < a.N; r += 256) {
      ld += log(a.S[(long)jo * a.strideS + (long)r * a.Np + r]);
      dt += a.alpha[(long)jo * a.Np + r] * a.yc[(long)jo * a.Np + r];
    }
    r1[tid] = ld;
    r2[tid] = dt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        r1[tid] += r1[tid + o];
        r2[tid] += r2[tid + o];
      }
      __syncthreads();
    }
    if (tid == 0) lml_s = 0.5 * (-(double)a.N * 1.8378770664093454836 - 2.0 * r1[0] - r2[0]);
    for (int t = tid >> 6; t < 2 + d; t += 4) {            // (a wave per component, four at a time)
      double sum = hypgrad_partial_sum(a.part, jo, a.nblk, 2 + d, t, tid & 63);
      if (t >= 2) sum /= a.hyp[jo].ls[t - 2];
      if ((tid & 63) == 0) res_s[t] = sum;
    }
    __syncthreads();
  }
  if (tid < P) {
#pragma clang fp contract(off)
    const int k = tid;
    const double* res = res_s;
    const double am1 = a.prior_a - 1.0;
    const double thk = th[k];
    const int fixed_k = fx[k];
    lpt[k] = a.prior_const + am1 * log(thk) - a.prior_b * thk;
    ljt[k] = fixed_k ? 0.0 : (thk > 36.0 ? thk : log(expm1(thk))) - thk;
    double g;
    if (k == 0) g = res[0];
    else if (k == P - 1) g = res[1];
    else if (nls == d) g = res[2 + (k - 1)];
    else {
      g = 0.0;
      for (int q = 0; q < d; ++q) g += res[2 + q];
    }
    const double em = expm1(thk);
    const double pg = (am1 / thk - a.prior_b) + (fixed_k ? 0.0 : 1.0 / em);
    tgt[k] = -(g + pg) * (thk > 36.0 ? 1.0 : -expm1(-thk));
  }
  __syncthreads();
  if (tid == 0) {
#pragma clang fp contract(off)
    bool bad = false;
    double lp = 0.0, lj = 0.0;
    for (int k = 0; k < P; ++k) lp += lpt[k];
    for (int k = 0; k < Pf; ++k) lj += ljt[kfree[k]];
    double obj = -lml_s - (lp + lj);
    for (int k = 0; k < Pf; ++k) {
      const double t = tgt[kfree[k]];
      tg[k] = t;
      bad = bad || !isfinite(t);
    }
    bad = bad || !isfinite(obj);
    if (bad) {
      obj = INFINITY;
      for (int k = 0; k < Pf; ++k) tg[k] = 0.0;
    }
    sc[0] = obj;
    a.n_eval[jo] += 1;
    if (mode == HS_POST) {
      if (bad) sc[3] = 1.0;
      const double h = -a.eps / 2.0;
      for (int k = 0; k < Pf; ++k) pm[k] += h * tg[k];
      if (it == a.iters - 1) {
        double pp = 0.0;
        for (int k = 0; k < Pf; ++k) pp += pm[k] * pm[k];
        const double H_new = sc[0] + Pf * half_log_2pi + pp / 2.0;
        const double kk = sc[2] > H_new ? 1.0 : exp(sc[2] - H_new);
        const bool diverged = sc[3] != 0.0;
        if (!diverged && isfinite(H_new) && a.uni[(long)jo * a.ns + i] < kk) {
          int kf = 0;
          for (int k = 0; k < P; ++k)
            if (!fx[k]) a.chains[((long)jo * a.ns + i) * P + kf++] = th[k];
          a.accepted[jo] += 1;
        } else {
          a.diverged[jo] += diverged ? 1 : 0;
          set_free_from_x(x_old);
          sc[0] = sc[1];
          for (int k = 0; k < Pf; ++k) tg[k] = tg_old[k];
        }
      }
    }
  }
}

int hmc_stream_state_doubles(int m) { return m * HS_STRIDE; }

void launch_hmc_stream(const HmcStreamArgs& a, int mode, int i, int it, hipStream_t s) {
  BOCF_LAUNCH(hmc_stream_kernel, dim3((unsigned)a.m), dim3(256), 0, s, a, mode, i, it);
}
