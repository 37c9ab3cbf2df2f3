// C ABI of libbocf_hip.so (declared in include/bocf_hip.h): context, memory, and the launch
// sequences of fit / predict / acquisition / selection.  No torch types, no CPU fallback.
#include "bocf_ctx.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <string>
#include <vector>

static thread_local std::string g_err;
int bocf_fail(const char* what, const char* detail) {
  g_err = std::string(what) + ": " + (detail ? detail : "");
  return -1;
}
void bocf_set_error(const char* text) { g_err = text ? text : ""; }

// first failed kernel launch since the last report (BOCF_LAUNCH, bocf_internal.h)
static thread_local std::string g_launch_err;
void bocf_note_launch(const char* kernel, hipError_t e) {
  if (e != hipSuccess && g_launch_err.empty()) g_launch_err = std::string("launch of ") + kernel + ": " + hipGetErrorString(e);
}
int bocf_launch_status() {
  const hipError_t e = hipGetLastError();
  if (!g_launch_err.empty()) {
    g_err = g_launch_err;
    g_launch_err.clear();
    return -1;
  }
  if (e != hipSuccess) return fail("hipGetLastError", hipGetErrorString(e));
  return 0;
}

extern "C" int bocf_version(void) { return 100; }
extern "C" const char* bocf_last_error(void) { return g_err.c_str(); }

extern "C" int bocf_create(int device, bocf_ctx** out) {
  if (!out) return fail("bocf_create", "null out");
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail("bocf_create", "no such HIP device");
  HIPCHK(hipSetDevice(device));
  bocf_ctx* c = new bocf_ctx();
  c->device = device;
  hipError_t e = hipStreamCreate(&c->stream);
  if (e != hipSuccess) {
    delete c;
    return fail("hipStreamCreate", hipGetErrorString(e));
  }
  {
    int lo_prio = 0, hi_prio = 0;   // the second stream carries the latency-critical side chains: highest priority
    (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
    e = hipStreamCreateWithPriority(&c->stream2, hipStreamDefault, hi_prio);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming);
  if (e != hipSuccess) {
    delete c;
    return fail("hipStreamCreate", hipGetErrorString(e));
  }
  *out = c;
  return 0;
}

static void drop_events(bocf_ctx* c) {
  for (auto& pr : c->events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  c->events.clear();
}

static void drop_phases(bocf_ctx* c) {
  for (auto& kv : c->phases)
    for (auto& pr : kv.second) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  c->phases.clear();
}

extern "C" void bocf_destroy(bocf_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)bocf_comm_destroy(c);
  if (c->shard_helper) bocf_destroy(c->shard_helper);
  c->shard_helper = nullptr;
  drop_events(c);
  drop_phases(c);
  if (c->pin_in) (void)hipHostFree(c->pin_in);
  if (c->pin_out) (void)hipHostFree(c->pin_out);
  if (c->fit_pin) (void)hipHostFree(c->fit_pin);
  if (c->up_pin) (void)hipHostFree(c->up_pin);
  if (c->ev_pin) (void)hipEventDestroy(c->ev_pin);
  DevBuf* bufs[] = {&c->R32, &c->Ri8, &c->Ri8e, &c->Ki8, &c->Ki8e, &c->X, &c->Xs, &c->S, &c->R, &c->RT, &c->E, &c->ET, &c->T, &c->yc, &c->tvec, &c->alpha, &c->lml, &c->jit, &c->hypd,
                    &c->info, &c->mu_train, &c->rvec, &c->dvec, &c->hmc_buf, &c->Xc, &c->Kstar, &c->meanpart, &c->sumsq, &c->mean, &c->var, &c->acq, &c->Vbuf, &c->dmean, &c->dvar, &c->dacq, &c->Vs, &c->Ws, &c->theta,
                    &c->prob, &c->best, &c->params, &c->Wt, &c->blk_idx, &c->blk_val, &c->out_idx, &c->out_val, &c->gpart, &c->gout, &c->pack, &c->gidx,
                    &c->gval, &c->shard_meta, &c->chol_flags};
  for (DevBuf* b : bufs) b->release();
  if (c->infer_out) (void)hipHostFree(c->infer_out);
  for (hipEvent_t ev : c->ev_parts) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : c->ev_chol) (void)hipEventDestroy(ev);
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  for (hipStream_t st : {c->s_res, c->s_hi, c->s_bulk, c->s_inv})
    if (st) (void)hipStreamDestroy(st);
  if (c->ev_half) (void)hipEventDestroy(c->ev_half);
  if (c->ev_inv_early) (void)hipEventDestroy(c->ev_inv_early);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

// ---------------------------------------------------------------------------------------------
// Options.  ONE table: name, accepted range, kind.  Kind 0 = speed only (schedules, tilings, workspace sizes: every setting
// computes the same result up to rounding, the tests pin that); kind 1 = documented semantics (fp32 contraction, the hyper-sample
// layout of the fitted outputs, data re-use between inferences: what the caller ASKS for); kind 2 = probes and test hooks
// (timing-only kernel variants whose results are wrong, the diagonal shift that forces the jitter ladder, the single-process
// stand-in for the ranks of a sharded fit) -- compiled only with -DBOCF_PROBES into libbocf_hip_probes.so, which tools/ and the
// tests that need a hook load; the product library has no such entry and rejects the names.  bocf_option_info / bocf_option_check
// need no GPU: the CPU suite enumerates the table (tests/test_host_cpu.py).
struct OptDesc {
  const char* name;
  long long lo, hi;
  int kind;
  void (*set)(bocf_ctx*, long long);
  bool (*extra)(long long);          // further constraint inside [lo, hi] (nullptr = none)
  const char* what;
};
static bool opt_lookahead_ok(long long v) { return v == -1 || v == 0 || v == 2; }
#ifdef BOCF_PROBES
static bool opt_swizzle_ok(long long v) { return v == -1 || v == 0 || v == 1 || (v >= 100 && v < 164) || (v >= 256 && v <= 258); }
static bool opt_potrf_ok(long long v) { return v == 0 || (v >= 11 && v <= 14); }
#else
static bool opt_swizzle_ok(long long v) { return v == -1 || v == 0 || v == 258; }
#endif
static const OptDesc g_options[] = {
    {"chunk", 128, 1 << 24, 0, [](bocf_ctx* c, long long v) { c->chunk = (long)round_up((int)v, 128); }, nullptr, "candidates per pass (rounded up to 128)"},
    {"workspace_mb", 1, 1 << 20, 0, [](bocf_ctx* c, long long v) { c->workspace_mb = (long)v; }, nullptr, "cap of the per-pass K* workspace"},
    {"profile", 0, 1, 0, [](bocf_ctx* c, long long v) { c->profile = v != 0; }, nullptr, "HIP events around the dominant kernel and the named phases"},
    {"predict_f32", 0, 1, 1, [](bocf_ctx* c, long long v) { c->predict_f32 = v != 0; }, nullptr, "fp32 variance contraction (BASELINE configs[4])"},
    {"i8_group", 0, 64, 0, [](bocf_ctx* c, long long v) { c->i8_group = (int)v; }, nullptr, "predict_i8: 0 = workgroups in blocks of 4 row-tile pairs x 8 column tiles per XCD; g >= 1 = bands of g pairs x all column tiles"},
    {"predict_i8", 0, 1, 1, [](bocf_ctx* c, long long v) { c->predict_i8 = v != 0; }, nullptr, "variance contraction in exact int8 products (six radix-254 digits per operand column, fp64 recombination)"},
    {"fused_infer", 0, 1, 0, [](bocf_ctx* c, long long v) { c->fused_infer = v != 0; }, nullptr, "one fused launch per inference for N <= 128"},
    {"reuse_data", 0, 1, 1, [](bocf_ctx* c, long long v) { c->reuse_data = v != 0; }, nullptr, "next fits reuse the resident X / Y"},
    {"skip_mu_train", 0, 1, 1, [](bocf_ctx* c, long long v) { c->skip_mu_train = v != 0; }, nullptr, "do not refresh the mean at the training inputs"},
    {"aggregate", 0, 8, 0, [](bocf_ctx* c, long long v) { c->aggregate = (int)v; }, nullptr, "panels per trailing update (0 = by size)"},
    {"lookahead", -1, 2, 0, [](bocf_ctx* c, long long v) { c->lookahead = (int)v; }, opt_lookahead_ok,
     "factorization schedule: -1 by size, 0 single stream, 2 reserved-CU chain"},
    {"team_fit", -1, 1, 0, [](bocf_ctx* c, long long v) { c->team_fit = (int)v; }, nullptr, "factorization + inverse in one launch by resident workgroup teams (-1 = by size)"},
    {"team_panels", 1, 32, 0, [](bocf_ctx* c, long long v) { c->team_panels = (int)v; }, nullptr, "team schedule above 8 panels: panels per team launch (one trailing update each)"},
    {"team_hybrid", 0, 2, 0, [](bocf_ctx* c, long long v) { c->team_hybrid = (int)v; }, nullptr, "more than 24 panels: launched schedule for the first block rows, one team launch for the rest"},
    {"team_tail_share", 1, 8, 0, [](bocf_ctx* c, long long v) { c->team_tail_share = (int)v; }, nullptr, "hybrid schedule: eighths of the compute units the tail's teams take"},
    {"team_whole_max", 2, 32, 0, [](bocf_ctx* c, long long v) { c->team_whole_max = (int)v; }, nullptr, "panels up to which one team launch does the whole factorization and inverse"},
    {"team_stream", 0, 1, 0, [](bocf_ctx* c, long long v) { c->team_stream = (int)v; }, nullptr, "teams: the critical tiles are formed underneath the diagonal blocks, 16 rows at a time"},
    {"team_crit_load", 0, 4096, 0, [](bocf_ctx* c, long long v) { c->team_crit_load = (int)v; }, nullptr, "teams: critical-chain workgroups carry nothing else while the others get by with <= this many units each"},
    {"lookahead_min_nb", 2, 1 << 20, 0, [](bocf_ctx* c, long long v) { c->lookahead_min_nb = (int)v; }, nullptr, "reserved-CU schedule from this many panels"},
    {"merge_x3", 0, 2, 0, [](bocf_ctx* c, long long v) { c->merge_x3 = (int)v; }, nullptr, "second product of an inverse merge in the three-buffer kernel"},
    {"shard_fit", 0, 1, 0, [](bocf_ctx* c, long long v) { c->shard_fit = v != 0; }, nullptr, "output-sharded fit over the communicator"},
    {"trsm_wave", 0, 1, 0, [](bocf_ctx* c, long long v) { c->trsm_wave = v != 0; }, nullptr, "row solves through the wave-level single-tile kernel"},
    {"overlap_inverse", -1, 1, 0, [](bocf_ctx* c, long long v) { c->overlap_inverse = (int)v; }, nullptr, "early part of the inverse underneath the factorization (-1 = by size)"},
    {"overlap", 0, 1, 0, [](bocf_ctx* c, long long v) { c->overlap = v != 0; }, nullptr, "K* build on a second stream"},
    {"small_path", 0, 1, 0, [](bocf_ctx* c, long long v) { c->small_path = v != 0; }, nullptr, "GEMV-shaped path for <= 16 candidates"},
    {"prefetch1", 0, 1, 0, [](bocf_ctx* c, long long v) { c->prefetch1 = v != 0; }, nullptr, "one-tile-deep staging in the 128-row variance kernel"},
    {"swizzle", -1, 258, 0, [](bocf_ctx* c, long long v) { c->swizzle = (int)v; }, opt_swizzle_ok, "variance-GEMM tiling: -1 by size, 0 128-row tiles, 258 256-row tiles (probes build: also 1, 100..163, 256, 257)"},
    {"hyper_samples", 1, 64, 1,
     [](bocf_ctx* c, long long v) {
       if ((int)v != c->hyper_samples) c->S_mc = 0;   // the transposed normals are laid out per group size
       c->hyper_samples = (int)v;
     },
     nullptr, "the fitted outputs are H hyper-samples x m / H model outputs"},
    {"acq_hyper_samples", 0, 64, 1, [](bocf_ctx* c, long long v) { c->acq_hyper_samples = (int)v; }, nullptr, "hyper-samples the acquisitions average over (0 = all)"},
    {"best_group", -1, 63, 1, [](bocf_ctx* c, long long v) { c->best_group = (int)v; }, nullptr, "whose best-so-far every hyper-sample uses (-1 = its own)"},
#ifdef BOCF_PROBES
    {"potrf_scalar", 0, 14, 2, [](bocf_ctx* c, long long v) { c->potrf_scalar = (int)v; }, opt_potrf_ok, "TIMING-ONLY variants of the diagonal-block kernel: 11..14 (wrong results)"},
    {"shard_fit_simulate", 0, 64, 2, [](bocf_ctx* c, long long v) { c->shard_fit_simulate = (int)v; }, nullptr, "TEST HOOK: one process plays all G ranks of a sharded fit"},
    {"kstar_valu_probe", 0, 4, 2, [](bocf_ctx* c, long long v) { c->kstar_valu_probe = (int)v; }, nullptr, "TIMING-ONLY variants of the two-buffer 256-row variance kernel (wrong results)"},
    {"test_diag_shift_1e12", -1000000000000000LL, 1000000000000000LL, 2, [](bocf_ctx* c, long long v) { c->test_diag_shift = (double)v * 1e-12; }, nullptr,
     "TEST HOOK: Ky diagonal -= value * 1e-12 (forces the jitter ladder)"},
    {"force_sched_timeout", 0, 1, 2, [](bocf_ctx* c, long long v) { c->force_sched_timeout = (int)v; }, nullptr, "TEST HOOK: the next gated schedule reports a dependency time-out"},
    {"force_cu_count", 0, 4096, 2, [](bocf_ctx* c, long long v) { c->force_cu_count = (int)v; }, nullptr, "TEST HOOK: pretend the device has this many compute units (schedule selection)"},
#endif
};
static const int g_noptions = (int)(sizeof(g_options) / sizeof(g_options[0]));

static const OptDesc* find_option(const char* name) {
  for (int i = 0; i < g_noptions; ++i)
    if (!strcmp(name, g_options[i].name)) return &g_options[i];
  return nullptr;
}

extern "C" int bocf_option_count(void) { return g_noptions; }

extern "C" int bocf_option_info(int index, const char** name_out, long long* lo_out, long long* hi_out, int* kind_out, const char** what_out) {
  if (index < 0 || index >= g_noptions) return fail("bocf_option_info", "index out of range");
  const OptDesc& o = g_options[index];
  if (name_out) *name_out = o.name;
  if (lo_out) *lo_out = o.lo;
  if (hi_out) *hi_out = o.hi;
  if (kind_out) *kind_out = o.kind;
  if (what_out) *what_out = o.what;
  return 0;
}

extern "C" int bocf_option_check(const char* name, long long value) {
  if (!name) return fail("bocf_set_option", "null argument");
  const OptDesc* o = find_option(name);
  if (!o) return fail("bocf_set_option", (std::string("unknown option '") + name + "'").c_str());
  if (value < o->lo || value > o->hi || (o->extra && !o->extra(value)))
    return fail("bocf_set_option", (std::string(name) + " = " + std::to_string(value) + " is out of range: " + o->what).c_str());
  return 0;
}

extern "C" int bocf_set_option(bocf_ctx* c, const char* name, long long value) {
  if (!c || !name) return fail("bocf_set_option", "null argument");
  if (bocf_option_check(name, value)) return -1;
  find_option(name)->set(c, value);
  return 0;
}

extern "C" int bocf_sync(bocf_ctx* c) {
  if (!c) return fail("bocf_sync", "null ctx");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// Test / inspection hook: a posterior handed over by the host instead of computed by fit + predict.
extern "C" int bocf_set_posterior(bocf_ctx* c, int m, int C, int N, const double* mean, const double* var, const double* mu_train) {
  if (!c || !mean || !var || !mu_train) return fail("bocf_set_posterior", "null argument");
  if (m < 1 || m > BOCF_MAX_FITS || C < 1 || N < 1) return fail("bocf_set_posterior", "m, C or N out of range");
  HIPCHK(hipSetDevice(c->device));
  const int cap = round_up(C, BOCF_TILE);
  if (c->mean.ensure(sizeof(double) * (size_t)m * cap) || c->var.ensure(sizeof(double) * (size_t)m * cap) || c->acq.ensure(sizeof(double) * cap) ||
      c->mu_train.ensure(sizeof(double) * (size_t)m * N))
    return -1;
  for (int j = 0; j < m; ++j) {
    HIPCHK(hipMemcpyAsync(c->mean.as<double>() + (size_t)j * cap, mean + (size_t)j * C, sizeof(double) * C, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->var.as<double>() + (size_t)j * cap, var + (size_t)j * C, sizeof(double) * C, hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(hipMemcpyAsync(c->mu_train.p, mu_train, sizeof(double) * (size_t)m * N, hipMemcpyHostToDevice, c->stream));
  c->mu_epoch++;
  HIPCHK(hipStreamSynchronize(c->stream));
  c->m = m; c->N = N; c->Np = round_up(N, BOCF_TILE); c->d = 1; c->C = C; c->pred_cap = cap;
  c->fitted = true;
  c->canned = true;
  c->have_acq = false;
  c->S_mc = 0;
  return 0;
}

#define BOCF_PIN_BYTES ((size_t)1 << 18)
static int pin_ensure(void** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap) return 0;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = bytes < 4096 ? 4096 : bytes;
  if (hipHostMalloc(p, want, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    *p = nullptr;
    return 1;                                              // (the caller takes the pageable path)
  }
  *cap = want;
  return 0;
}

extern "C" int bocf_set_candidates(bocf_ctx* c, const double* Xc, int C) {
  if (!c || !c->fitted) return fail("bocf_set_candidates", "model not fitted");
  if (c->canned) return fail("bocf_set_candidates", "the context holds a host-given posterior (bocf_set_posterior): fit first");
  if (C < 0 || (C > 0 && !Xc)) return fail("bocf_set_candidates", "bad candidate batch");
  HIPCHK(hipSetDevice(c->device));
  c->have_acq = false;
  c->C = C;
  if (C == 0) return 0;
  if (c->Xc.ensure(sizeof(double) * (size_t)C * c->d)) return -1;
  const size_t bytes = sizeof(double) * (size_t)C * c->d;
  if (bytes <= BOCF_PIN_BYTES && pin_ensure(&c->pin_in, &c->pin_in_cap, bytes) == 0) {
    // small batch: through the pinned buffer, no synchronisation here (whatever reads the candidates is ordered behind the copy on the stream)
    if (!c->ev_pin) HIPCHK(hipEventCreateWithFlags(&c->ev_pin, hipEventDisableTiming));
    else HIPCHK(hipEventSynchronize(c->ev_pin));          // (an upload still reading the buffer: only when nothing synchronised in between)
    memcpy(c->pin_in, Xc, bytes);
    HIPCHK(hipMemcpyAsync(c->Xc.p, c->pin_in, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev_pin, c->stream));
    return 0;
  }
  HIPCHK(hipMemcpyAsync(c->Xc.p, Xc, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// mean / var of all resident candidates into c->mean / c->var (m, pred_cap)
static int run_predict(bocf_ctx* c, int flags, bool need_var, bool need_grad = false) {
  const int N = c->N, Np = c->Np, m = c->m, d = c->d, C = c->C;
  if (C == 0) return 0;
  if (c->canned) {           // mean / var were given by the host (bocf_set_posterior): nothing to predict
    if (need_grad) return fail("predict", "a host-given posterior carries no gradients");
    return 0;
  }
  const int nrt = Np / BOCF_TILE;
  // candidates per pass: option "chunk", lowered so that the K* (and, for gradients, V) workspace of ALL fitted outputs
  // (hyper-samples x outputs) stays inside option "workspace_mb"; results do not depend on the chunking
  long chunk = c->chunk;
  {
    const double per_col = (double)m * Np * sizeof(double) * (need_grad ? 2.0 : 1.0);
    long fit_cols = (long)((double)c->workspace_mb * 1048576.0 / per_col);
    fit_cols = fit_cols / BOCF_TILE * BOCF_TILE;
    if (fit_cols < BOCF_TILE) fit_cols = BOCF_TILE;
    if (chunk > fit_cols) chunk = fit_cols;
  }
  const int chunkpad = (int)(C < chunk ? round_up(C, BOCF_TILE) : chunk);
  if (c->pred_cap < C) {
    const int cap = round_up(C, BOCF_TILE);
    if (c->mean.ensure(sizeof(double) * (size_t)m * cap) || c->var.ensure(sizeof(double) * (size_t)m * cap) ||
        c->acq.ensure(sizeof(double) * cap))
      return -1;
    c->pred_cap = cap;
  }
  // pred_cap may have been sized for another m: keep the leading dimension explicit
  const long ld = c->pred_cap;
  if (c->mean.ensure(sizeof(double) * (size_t)m * ld) || c->var.ensure(sizeof(double) * (size_t)m * ld)) return -1;
  if (need_var) {
    // (the small path keeps one partial per 16-row tile)
    if (c->Kstar.ensure(sizeof(double) * (size_t)m * Np * chunkpad) || c->sumsq.ensure(sizeof(double) * (size_t)m * (C <= BOCF_SMALL_N ? Np / 16 : nrt) * chunkpad)) return -1;
  }
  const bool small = C <= BOCF_SMALL_N && c->small_path;
  // fp32 variance contraction (BASELINE configs[4]): K* stored as fp32, R32 = (float) R; the fit, the mean
  // (whose alpha-weighted sum cancels catastrophically in fp32) and the gradient path stay in fp64
  const bool f32 = c->predict_f32 && need_var && !need_grad && !small;
  if (f32 && !c->r32_valid) {
    if (c->R32.ensure(sizeof(float) * (size_t)m * Np * Np)) return -1;
    launch_f64_to_f32(c->R.as<double>(), c->R32.as<float>(), (long)m * Np * Np, c->stream);
    c->r32_valid = true;
  }
  // int8 (Ozaki) contraction: from 128 candidates up, variances only (the gradient path needs V itself)
  const bool i8 = c->predict_i8 && need_var && !need_grad && !small && !f32 && Np <= 16384;   // (int32 group sums: 6 x 127^2 x N < 2^31)
  if (i8) {
    if (c->Ki8.ensure(i8_operand_bytes(Np, chunkpad, m)) || c->Ki8e.ensure(sizeof(int) * m)) return -1;
    if (!c->ri8_valid) {
      if (c->Ri8.ensure(i8_operand_bytes(Np, Np, m)) || c->Ri8e.ensure(sizeof(int) * (size_t)m * Np)) return -1;
      HIPCHK(hipMemsetAsync(c->Ri8e.p, 0x80, sizeof(int) * (size_t)m * Np, c->stream));      // (below any exponent: the kernel takes maxima)
      launch_col_exponents(c->R.as<double>(), (long)Np * Np, Np, c->Ri8e.as<int>(), m, c->stream);
      launch_slice_operand(c->R.as<double>(), Np, (long)Np * Np, Np, Np, Np, c->Ri8e.as<int>(), Np, c->Ri8.p, m, c->stream);
      // a stationary kernel never exceeds its variance: ONE scale for every column of K*
      std::vector<int> eb(m);
      for (int j = 0; j < m; ++j) eb[j] = c->hyp[j].variance > 0.0 ? ilogb(c->hyp[j].variance) + 1 : 0;
      HIPCHK(hipMemcpyAsync(c->Ki8e.p, eb.data(), sizeof(int) * m, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));             // (eb goes out of scope)
      c->ri8_valid = true;
    }
  }
  if (small && need_var) {
    if (c->Vs.ensure(sizeof(double) * (size_t)m * Np * BOCF_SMALL_N) || c->Ws.ensure(sizeof(double) * (size_t)m * Np * BOCF_SMALL_N)) return -1;
  }
  if (need_grad) {
    if ((!small && c->Vbuf.ensure(sizeof(double) * (size_t)m * Np * chunkpad)) || c->dmean.ensure(sizeof(double) * (size_t)m * ld * d) ||
        c->dvar.ensure(sizeof(double) * (size_t)m * ld * d) || c->dacq.ensure(sizeof(double) * (size_t)ld * d))
      return -1;
  }
  const size_t mean_plane = (size_t)m * nrt * (chunkpad > Np ? chunkpad : Np);     // partial means per 128-row block: hi plane, lo plane
  if (c->meanpart.ensure(sizeof(double) * 2 * mean_plane)) return -1;
  const long strideS = (long)Np * Np;
  for (long c0 = 0; c0 < C; c0 += chunk) {
    const int Cn = (int)((C - c0) < chunk ? (C - c0) : chunk);
    const int Cpad = round_up(Cn, BOCF_TILE);
    // The chunk is processed in up to 4 column parts: part i's K* build (VALU + HBM writes, stream2)
    // runs underneath part i-1's contraction (MFMA, main stream).  Parts share the chunk's buffers
    // (disjoint column ranges), and per-candidate results do not depend on the partition.
    int nparts = 1;
    if (c->overlap && need_var && !(C <= BOCF_SMALL_N && c->small_path)) nparts = Cpad >= 32768 ? 4 : (Cpad >= 8192 ? 2 : 1);
    if (nparts > 1) {
      while ((int)c->ev_parts.size() < nparts) {
        hipEvent_t ev;
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c->ev_parts.push_back(ev);
      }
      HIPCHK(hipEventRecord(c->ev_start, c->stream));
      HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_start, 0));
    }
    const int part_cols = round_up((Cpad + nparts - 1) / nparts, BOCF_TILE);
    for (int part = 0; part < nparts; ++part) {
      const int pc0 = part * part_cols;                       // first column of the part inside the chunk
      if (pc0 >= Cpad) break;
      const int pcols = (pc0 + part_cols <= Cpad) ? part_cols : (Cpad - pc0);
      const int pvalid = Cn - pc0 < 0 ? 0 : (Cn - pc0 < pcols ? Cn - pc0 : pcols);
      hipStream_t sx = nparts > 1 ? c->stream2 : c->stream;
      const int ns = nsplit_for(Np, pcols, m);
      double* kbase = f32 ? reinterpret_cast<double*>(c->Kstar.as<float>() + pc0) : c->Kstar.as<double>() + pc0;
      PhaseTimer t_cross(c, nparts > 1 ? "cross_overlapped" : "cross");
      if (nparts > 1) t_cross.stop();        // (events belong to the main stream; the overlapped build runs on stream2)
      // (<= 16 candidates with variances: the small path on the matrix pipe -- K* and the same mean partials from cross_small_kernel)
      const bool small_mfma = small && need_var && !f32;
      int nc_small = 1;
      while (nc_small < Cn) nc_small *= 2;
      if (small_mfma)
        launch_cross_small(c->Xs.as<double>(), c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->Xc.as<double>(), (int)c0, Cn, nc_small,
                           c->alpha.as<double>(), c->Kstar.as<double>(), Cpad, (long)Np * Cpad, c->meanpart.as<double>(),
                           c->meanpart.as<double>() + mean_plane, pcols, m, sx, BOCF_KIDS(c));
      else
      launch_cross_kernel(c->Xs.as<double>(), c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->Xc.as<double>(),
                          (int)c0 + pc0, pvalid, pcols, c->alpha.as<double>(), kbase, Cpad, (long)Np * Cpad,
                          c->meanpart.as<double>() + (size_t)pc0 * m * nrt, c->meanpart.as<double>() + mean_plane + (size_t)pc0 * m * nrt, ns, m,
                          need_var ? (f32 ? 2 : 1) : 0, sx, BOCF_KIDS(c));
      // (a part that goes on to the big contraction on the same stream finishes its means in the launch that finishes its variances)
      const bool mean_with_var = need_var && nparts == 1 && (!small || small_mfma);
      if (!mean_with_var)
        launch_finalize_mean(c->meanpart.as<double>() + (size_t)pc0 * m * nrt, c->meanpart.as<double>() + mean_plane + (size_t)pc0 * m * nrt, nrt,
                             pcols, c->hypd.as<KernHyp>(), c->mean.as<double>(), ld, (int)c0 + pc0, pvalid, m, sx);
      const double* mp_hi = mean_with_var ? c->meanpart.as<double>() + (size_t)pc0 * m * nrt : nullptr;
      const double* mp_lo = mean_with_var ? c->meanpart.as<double>() + mean_plane + (size_t)pc0 * m * nrt : nullptr;
      t_cross.stop();
      if (!need_var) continue;
      if (nparts > 1) {
        HIPCHK(hipEventRecord(c->ev_parts[part], c->stream2));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_parts[part], 0));
      }
      if (small) {
        // n <= 16: GEMV-shaped, R streamed once per product (single-point L-BFGS calls)
        const int nc = nc_small;
        if (small_mfma) {
          launch_gemv_small_t_mfma(c->R.as<double>(), strideS, Np, c->Kstar.as<double>(), Cpad, (long)Np * Cpad, c->Vs.as<double>(), c->sumsq.as<double>(), Cpad,
                                   nc, m, c->stream);
          launch_finalize_var(c->sumsq.as<double>(), Np / 16, Cpad, c->hypd.as<KernHyp>(), flags, c->var.as<double>(), ld, (int)c0, Cn, m, c->stream, mp_hi,
                              mp_lo, nrt, c->mean.as<double>());
        } else {
          launch_gemv_small_t(c->R.as<double>(), strideS, Np, c->Kstar.as<double>(), Cpad, (long)Np * Cpad, c->Vs.as<double>(), nc, m, c->stream);
          launch_sumsq_small(c->Vs.as<double>(), Np, c->sumsq.as<double>(), Cpad, nc, m, c->stream);
          launch_finalize_var(c->sumsq.as<double>(), 1, Cpad, c->hypd.as<KernHyp>(), flags, c->var.as<double>(), ld, (int)c0, Cn, m, c->stream);
        }
        if (need_grad) {
          if (small_mfma) launch_gemv_small_n_mfma(c->RT.as<double>(), strideS, Np, c->Vs.as<double>(), c->Ws.as<double>(), nc, m, c->stream);
          else launch_gemv_small_n(c->R.as<double>(), strideS, Np, c->Vs.as<double>(), c->Ws.as<double>(), nc, m, c->stream);
          launch_grad_kernel(c->Xs.as<double>(), c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->Xc.as<double>(), (int)c0, Cn,
                             c->alpha.as<double>(), c->Ws.as<double>(), nc, (long)Np * nc, c->dmean.as<double>(),
                             c->dvar.as<double>(), ld, m, c->stream, BOCF_KIDS(c));
        }
        continue;
      }
      if (f32) {
        GemmArgs32 g32{};
        g32.A = c->R32.as<float>(); g32.lda = Np; g32.strideA = strideS;
        g32.B = c->Kstar.as<float>() + pc0; g32.ldb = Cpad; g32.strideB = (long)Np * Cpad;
        g32.M = Np; g32.Ncols = pcols; g32.K = Np;
        g32.sumsq = c->sumsq.as<double>() + (size_t)pc0 * m * nrt; g32.strideSumsq = (long)nrt * pcols;
        g32.tile128 = c->swizzle == 0;
        hipEvent_t f0 = nullptr, f1 = nullptr;
        if (c->profile) {
          HIPCHK(hipEventCreate(&f0));
          HIPCHK(hipEventCreate(&f1));
          HIPCHK(hipEventRecord(f0, c->stream));
        }
        launch_gemm_f32_sumsq(g32, m, c->stream);
        if (c->profile) {
          HIPCHK(hipEventRecord(f1, c->stream));
          c->events.emplace_back(f0, f1);
          c->prof_flops += (double)m * (double)N * (double)N * (double)pvalid;
        }
        launch_finalize_var(c->sumsq.as<double>() + (size_t)pc0 * m * nrt, nrt, pcols, c->hypd.as<KernHyp>(), flags, c->var.as<double>(), ld,
                            (int)c0 + pc0, pvalid, m, c->stream, mp_hi, mp_lo, nrt, c->mean.as<double>());
        continue;
      }
      if (i8) {
        // the part's K* -> digit fragments, then the exact int8 contraction; same partial sums' layout, same finalisation
        hipEvent_t f0 = nullptr, f1 = nullptr;
        if (c->profile) {
          HIPCHK(hipEventCreate(&f0));
          HIPCHK(hipEventCreate(&f1));
          HIPCHK(hipEventRecord(f0, c->stream));
        }
        launch_slice_operand(c->Kstar.as<double>() + pc0, Cpad, (long)Np * Cpad, Np, Np, pcols, c->Ki8e.as<int>(), 0, c->Ki8.p, m, c->stream);
        launch_var_i8(c->Ri8.p, c->Ki8.p, Np, pcols, c->Ri8e.as<int>(), c->Ki8e.as<int>(), c->sumsq.as<double>() + (size_t)pc0 * m * nrt, (long)nrt * pcols, m,
                      c->stream, c->i8_group);
        if (c->profile) {
          HIPCHK(hipEventRecord(f1, c->stream));
          c->events.emplace_back(f0, f1);
          c->prof_flops += (double)m * (double)N * (double)N * (double)pvalid;
        }
        launch_finalize_var(c->sumsq.as<double>() + (size_t)pc0 * m * nrt, nrt, pcols, c->hypd.as<KernHyp>(), flags, c->var.as<double>(), ld,
                            (int)c0 + pc0, pvalid, m, c->stream, mp_hi, mp_lo, nrt, c->mean.as<double>());
        continue;
      }
      // V = R^T K*, only its column sums of squares leave the chip
      GemmArgs g{};
      g.A = c->R.as<double>(); g.lda = Np; g.strideA = strideS;
      g.B = c->Kstar.as<double>() + pc0; g.ldb = Cpad; g.strideB = (long)Np * Cpad;
      g.M = Np; g.Ncols = pcols; g.K = Np; g.kb = BOCF_TILE; g.krt = BOCF_TILE; g.rt_desc = 1;
      // 256-row tiles (two 128-row tiles per workgroup share every K* fetch: 77.6 instead of 144 GB per launch at config 3,
      // bit-identical sums) in the three-buffer kernel whose loop keeps the vector ALU free and skips the zero blocks of R's
      // diagonal range (gemm_f64.hip): 0.93 of the fp64 MFMA peak against 0.83 for the 128-row kernel at N = 4096, 0.81 against
      // 0.77 at config 2 (N = 1024, 8192 candidates); from 2048 candidates per pass up (below that its fewer, larger
      // workgroups leave CUs idle: N = 1024, C = 1024: 0.49 against 0.40 ms for the 128-row kernel).  Padded sizes that are not
      // a multiple of 256 fall back in the launcher.  Option "swizzle" = 0 / 256 / 257 / 258 forces a tiling.
      g.swizzle = c->swizzle < 0 ? (pcols >= 2048 ? 258 : 0) : c->swizzle;
      g.vprobe = c->kstar_valu_probe;
      g.prefetch1 = c->prefetch1 || nparts > 1;     // 194 VGPRs: leaves room for the K*-build waves on the same SIMD
      g.sumsq = c->sumsq.as<double>() + (size_t)pc0 * m * nrt; g.strideSumsq = (long)nrt * pcols;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (c->profile) {
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, c->stream));
      }
      launch_gemm_f64(g, m, 1, c->stream);
      if (c->profile) {
        HIPCHK(hipEventRecord(e1, c->stream));
        c->events.emplace_back(e0, e1);
        c->prof_flops += (double)m * (double)N * (double)N * (double)pvalid;
      }
      launch_finalize_var(c->sumsq.as<double>() + (size_t)pc0 * m * nrt, nrt, pcols, c->hypd.as<KernHyp>(), flags, c->var.as<double>(), ld,
                          (int)c0 + pc0, pvalid, m, c->stream, mp_hi, mp_lo, nrt, c->mean.as<double>());
    }
    if (small || !need_var) continue;
    if (!need_grad) continue;
    // gradients need w = Ky^-1 k* = R (R^T k*): V = R^T K* stored this time, then W = R V (R k-major = RT);
    // W overwrites the K* buffer (no longer needed: the gradient kernel recomputes dk/dx from the inputs)
    GemmArgs v{};
    v.A = c->R.as<double>(); v.lda = Np; v.strideA = strideS;
    v.B = c->Kstar.as<double>(); v.ldb = Cpad; v.strideB = (long)Np * Cpad;
    v.Cin = nullptr; v.Cout = c->Vbuf.as<double>(); v.ldc = Cpad; v.strideC = (long)Np * Cpad;
    v.M = Np; v.Ncols = Cpad; v.K = Np; v.kb = BOCF_TILE; v.krt = BOCF_TILE; v.rt_desc = 1; v.alpha = 1.0;
    launch_gemm_f64(v, m, 0, c->stream);
    GemmArgs w{};
    w.A = c->RT.as<double>(); w.lda = Np; w.strideA = strideS;
    w.B = c->Vbuf.as<double>(); w.ldb = Cpad; w.strideB = (long)Np * Cpad;
    w.Cin = nullptr; w.Cout = c->Kstar.as<double>(); w.ldc = Cpad; w.strideC = (long)Np * Cpad;
    w.M = Np; w.Ncols = Cpad; w.K = Np; w.kb = Np; w.kbeg_rt = BOCF_TILE; w.alpha = 1.0;
    launch_gemm_f64(w, m, 0, c->stream);
    launch_grad_kernel(c->Xs.as<double>(), c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->Xc.as<double>(), (int)c0, Cn,
                       c->alpha.as<double>(), c->Kstar.as<double>(), Cpad, (long)Np * Cpad, c->dmean.as<double>(), c->dvar.as<double>(), ld,
                       m, c->stream, BOCF_KIDS(c));
  }
  LAUNCHCHK();
  return 0;
}

// `rows` rows of `len` doubles (device row stride ld) to contiguous host rows, for up to four arrays, then ONE stream synchronisation; small
// totals go through the pinned buffer (a copy into pageable memory is staged and waited for one by one)
struct RowsOut { const double* dev; long ld; int rows; size_t len; double* out; };
static int copy_rows_out_sync(bocf_ctx* c, const RowsOut* v, int n) {
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += v[i].out ? sizeof(double) * v[i].rows * v[i].len : 0;
  const bool pinned = total > 0 && total <= BOCF_PIN_BYTES && pin_ensure(&c->pin_out, &c->pin_out_cap, total) == 0;
  char* pin = static_cast<char*>(c->pin_out);
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    if (!v[i].out) continue;
    for (int j = 0; j < v[i].rows; ++j) {
      void* dst = pinned ? static_cast<void*>(pin + off) : static_cast<void*>(v[i].out + (size_t)j * v[i].len);
      HIPCHK(hipMemcpyAsync(dst, v[i].dev + (size_t)j * v[i].ld, sizeof(double) * v[i].len, hipMemcpyDeviceToHost, c->stream));
      off += sizeof(double) * v[i].len;
    }
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (pinned) {
    off = 0;
    for (int i = 0; i < n; ++i) {
      if (!v[i].out) continue;
      const size_t b = sizeof(double) * v[i].rows * v[i].len;
      memcpy(v[i].out, pin + off, b);
      off += b;
    }
  }
  return 0;
}

extern "C" int bocf_predict(bocf_ctx* c, int flags, double* mean_out, double* var_out) {
  if (!c || !c->fitted || c->canned) return fail("bocf_predict", "model not fitted");
  HIPCHK(hipSetDevice(c->device));
  if (c->C == 0) return 0;
  if (run_predict(c, flags, var_out != nullptr)) return -1;
  const RowsOut v[2] = {{c->mean.as<double>(), c->pred_cap, c->m, (size_t)c->C, mean_out}, {c->var.as<double>(), c->pred_cap, c->m, (size_t)c->C, var_out}};
  return copy_rows_out_sync(c, v, 2);
}

// multi_outputGP.predict(X, full_cov=True): see include/bocf_hip.h.  Three steps on kernels the mean path already has: k0 = K(X, x_0) (the
// cross kernel on the one candidate, stored), w = R (R^T k0) (the two GEMVs of the alpha solve), then the ordinary mean pass over all the
// candidates with w in the place of alpha, and one finalisation kernel.
extern "C" int bocf_predict_cov_column(bocf_ctx* c, int flags, double* cov_out) {
  if (!c || !c->fitted || c->canned) return fail("bocf_predict_cov_column", "model not fitted");
  if (!cov_out) return fail("bocf_predict_cov_column", "null output");
  HIPCHK(hipSetDevice(c->device));
  if (c->C == 0) return 0;
  const int N = c->N, Np = c->Np, m = c->m, d = c->d, nrt = Np / BOCF_TILE;
  const long strideS = (long)Np * Np;
  const size_t plane = (size_t)m * nrt * (Np > BOCF_TILE ? Np : BOCF_TILE);
  if (c->Kstar.ensure(sizeof(double) * (size_t)m * Np * BOCF_TILE) || c->meanpart.ensure(sizeof(double) * 2 * plane) ||
      c->tvec.ensure(sizeof(double) * (size_t)m * Np) || c->dvec.ensure(sizeof(double) * (size_t)m * Np))
    return -1;
  launch_cross_kernel(c->Xs.as<double>(), c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->Xc.as<double>(), 0, 1, BOCF_TILE,
                      c->alpha.as<double>(), c->Kstar.as<double>(), BOCF_TILE, (long)Np * BOCF_TILE, c->meanpart.as<double>(),
                      c->meanpart.as<double>() + plane, 1, m, 1, c->stream, BOCF_KIDS(c));
  launch_gemv_small_t(c->R.as<double>(), strideS, Np, c->Kstar.as<double>(), BOCF_TILE, (long)Np * BOCF_TILE, c->tvec.as<double>(), 1, m, c->stream);
  launch_gemv_upper_n(c->R.as<double>(), strideS, Np, c->tvec.as<double>(), c->dvec.as<double>(), m, c->stream);
  // the mean pass reads c->alpha: lend it w for one pass (c->mean then holds K(x_i, X) w + ymean)
  std::swap(c->alpha.p, c->dvec.p);
  std::swap(c->alpha.cap, c->dvec.cap);
  const int rc = run_predict(c, 0, false);
  std::swap(c->alpha.p, c->dvec.p);
  std::swap(c->alpha.cap, c->dvec.cap);
  if (rc) return -1;
  if (c->var.ensure(sizeof(double) * (size_t)m * c->pred_cap)) return -1;
  launch_cov_column(c->Xc.as<double>(), c->C, d, c->kernel_id, c->hypd.as<KernHyp>(), c->mean.as<double>(), c->pred_cap, flags, c->var.as<double>(),
                    c->pred_cap, m, c->stream, BOCF_KIDS(c));
  const RowsOut v[1] = {{c->var.as<double>(), c->pred_cap, m, (size_t)c->C, cov_out}};
  if (copy_rows_out_sync(c, v, 1)) return -1;
  LAUNCHCHK();
  c->have_acq = false;                                       // c->mean / c->var no longer hold the posterior the acquisition kernels read
  return 0;
}

extern "C" int bocf_predict_gradients(bocf_ctx* c, double* dmean_out, double* dvar_out) {
  if (!c || !c->fitted || c->canned) return fail("bocf_predict_gradients", "model not fitted");
  HIPCHK(hipSetDevice(c->device));
  if (c->C == 0) return 0;
  if (run_predict(c, BOCF_ADD_NOISE | BOCF_CLIP, true, true)) return -1;
  const RowsOut v[2] = {{c->dmean.as<double>(), (long)c->pred_cap * c->d, c->m, (size_t)c->C * c->d, dmean_out},
                        {c->dvar.as<double>(), (long)c->pred_cap * c->d, c->m, (size_t)c->C * c->d, dvar_out}};
  if (copy_rows_out_sync(c, v, 2)) return -1;
  LAUNCHCHK();
  return 0;
}

extern "C" int bocf_mean_at_train(bocf_ctx* c, double* out) {
  if (!c || !c->fitted || !out) return fail("bocf_mean_at_train", "model not fitted / null out");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(out, c->mu_train.p, sizeof(double) * (size_t)c->m * c->N, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

static int upload_acq_params(bocf_ctx* c, const double* theta, int theta_dim, const double* prob, int L, const double* params, int nparams) {
  if (L < 1 || L > BOCF_MAX_L) return fail("acquisition", "L out of range (1..32)");
  std::vector<double> th((size_t)L * (theta_dim > 0 ? theta_dim : 1), 0.0), pr(L), pa(BOCF_MAX_M, 0.0);
  if (theta_dim > 0) {
    if (!theta) return fail("acquisition", "theta is null");
    memcpy(th.data(), theta, sizeof(double) * (size_t)L * theta_dim);
  }
  for (int l = 0; l < L; ++l) pr[l] = prob ? prob[l] : 1.0 / L;   // maEI.py:50 vs :52
  if (nparams > BOCF_MAX_M) return fail("acquisition", "too many utility parameters");
  for (int i = 0; i < nparams; ++i) pa[i] = params[i];
  // L-BFGS refinement calls the acquisition hundreds of times with the same parameters: upload only on change
  std::vector<double> key;
  key.reserve(th.size() + pr.size() + pa.size() + 2);
  key.push_back((double)L);
  key.push_back((double)theta_dim);
  key.insert(key.end(), th.begin(), th.end());
  key.insert(key.end(), pr.begin(), pr.end());
  key.insert(key.end(), pa.begin(), pa.end());
  if (key.size() == c->last_params.size() && !memcmp(key.data(), c->last_params.data(), sizeof(double) * key.size())) return 0;
  if (c->theta.ensure(sizeof(double) * th.size()) || c->prob.ensure(sizeof(double) * L) || c->best.ensure(sizeof(double) * L) ||
      c->params.ensure(sizeof(double) * BOCF_MAX_M))
    return -1;
  HIPCHK(hipMemcpyAsync(c->theta.p, th.data(), sizeof(double) * th.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->prob.p, pr.data(), sizeof(double) * L, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->params.p, pa.data(), sizeof(double) * BOCF_MAX_M, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->last_params.swap(key);
  c->best_epoch = -1;                                    // the best-so-far values belong to the old parameters
  return 0;
}

// outputs per hyper-sample; -1 (with the error set) when the fit does not hold H whole groups
static int group_size(bocf_ctx* c, const char* where) {
  const int H = c->hyper_samples;
  if (c->m % H != 0) return fail(where, "the fitted outputs are not a multiple of option hyper_samples"), -1;
  if (c->best_group >= H) return fail(where, "option best_group is not a valid hyper-sample index"), -1;
  if (c->m / H > BOCF_MAX_M) return fail(where, "too many outputs per hyper-sample"), -1;
  return c->m / H;
}

// acquisition values (and gradients) to the host: small batches through the pinned buffer -- both copies asynchronous, one synchronisation
static int finish_acq(bocf_ctx* c, double* acq_out, double* dacq_out = nullptr) {
  c->have_acq = true;
  const size_t b0 = acq_out ? sizeof(double) * (size_t)c->C : 0, b1 = dacq_out ? sizeof(double) * (size_t)c->C * c->d : 0;
  if (b0 + b1 > 0 && b0 + b1 <= BOCF_PIN_BYTES && pin_ensure(&c->pin_out, &c->pin_out_cap, b0 + b1) == 0) {
    char* pin = static_cast<char*>(c->pin_out);
    if (b0) HIPCHK(hipMemcpyAsync(pin, c->acq.p, b0, hipMemcpyDeviceToHost, c->stream));
    if (b1) HIPCHK(hipMemcpyAsync(pin + b0, c->dacq.p, b1, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (b0) memcpy(acq_out, pin, b0);
    if (b1) memcpy(dacq_out, pin + b0, b1);
    LAUNCHCHK();
    return 0;
  }
  if (dacq_out) HIPCHK(hipMemcpyAsync(dacq_out, c->dacq.p, b1, hipMemcpyDeviceToHost, c->stream));
  if (acq_out) HIPCHK(hipMemcpyAsync(acq_out, c->acq.p, b0, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  LAUNCHCHK();
  return 0;
}

// The reference's h-loop (maEI.py:85-97, uEI_noiseless.py:71-82) runs here: hyper-sample h reads rows
// [h*m, (h+1)*m) of the mean / variance / gradient buffers and adds its share (1/H) to acq (and dacq).
template <typename Launch>
static int acq_over_hyper_samples(bocf_ctx* c, AcqArgs a, int m, int linear, Launch launch) {
  const int H = (c->acq_hyper_samples > 0 && c->acq_hyper_samples < c->hyper_samples) ? c->acq_hyper_samples : c->hyper_samples;
  const double* mean = a.mean; const double* var = a.var; const double* dmean = a.dmean; const double* dvar = a.dvar;
  for (int h = 0; h < H; ++h) {
    if (h == 0 || c->best_group < 0) {
      const int gb = c->best_group >= 0 ? c->best_group : h;
      // one hyper-sample: best-so-far depends on the train mean and the parameters only -- the hundreds of acquisition calls of one
      // BO step (batch call + L-BFGS refinements) share it
      const int sig = (linear ? 1 : 0) | (a.util_kind << 1) | (gb << 8);
      const bool cached = H == 1 && c->best_epoch == c->mu_epoch && c->best_sig == sig;
      c->best_epoch = H == 1 ? c->mu_epoch : -1;
      c->best_sig = sig;
      if (!cached)
      launch_best_so_far(c->mu_train.as<double>() + (size_t)gb * m * c->N, c->N, m, linear, a.util_kind, a.theta, a.theta_dim, a.L,
                         a.util_params, c->best.as<double>(), c->stream);
    }
    a.mean = mean + (size_t)h * m * a.ld;
    a.var = var + (size_t)h * m * a.ld;
    if (dmean) { a.dmean = dmean + (size_t)h * m * a.ldg * a.d; a.dvar = dvar + (size_t)h * m * a.ldg * a.d; }
    a.accumulate = h > 0;
    a.scale = 1.0 / H;
    PhaseTimer t(c, "acq");
    launch(a, c->stream);
  }
  return 0;
}

extern "C" int bocf_acq_linear(bocf_ctx* c, int kind, const double* theta, const double* prob, int L, double* acq_out) {
  if (!c || !c->fitted) return fail("bocf_acq_linear", "model not fitted");
  if (kind != BOCF_ACQ_EI && kind != BOCF_ACQ_PI) return fail("bocf_acq_linear", "unknown acquisition kind");
  const int m = group_size(c, "bocf_acq_linear");
  if (m < 0) return -1;
  HIPCHK(hipSetDevice(c->device));
  if (c->C == 0) return 0;
  if (upload_acq_params(c, theta, m, prob, L, nullptr, 0)) return -1;
  if (run_predict(c, BOCF_ADD_NOISE | BOCF_CLIP, true)) return -1;        // model.predict (maEI.py:87)
  AcqArgs a{};
  a.mean = c->mean.as<double>(); a.var = c->var.as<double>(); a.ld = c->pred_cap;
  a.m = m; a.C = c->C; a.L = L; a.kind = kind; a.util_kind = BOCF_UTIL_LINEAR; a.theta_dim = m;
  a.theta = c->theta.as<double>(); a.prob = c->prob.as<double>(); a.best = c->best.as<double>();
  a.util_params = c->params.as<double>(); a.acq = c->acq.as<double>();
  if (acq_over_hyper_samples(c, a, m, 1, launch_acq_linear)) return -1;
  return finish_acq(c, acq_out);
}

static int finish_acq_grad(bocf_ctx* c, double* acq_out, double* dacq_out) { return finish_acq(c, acq_out, dacq_out); }

extern "C" int bocf_acq_linear_grad(bocf_ctx* c, int kind, const double* theta, const double* prob, int L, double* acq_out,
                                    double* dacq_out) {
  if (!c || !c->fitted) return fail("bocf_acq_linear_grad", "model not fitted");
  if (kind != BOCF_ACQ_EI && kind != BOCF_ACQ_PI) return fail("bocf_acq_linear_grad", "unknown acquisition kind");
  const int m = group_size(c, "bocf_acq_linear_grad");
  if (m < 0) return -1;
  HIPCHK(hipSetDevice(c->device));
  if (c->C == 0) return 0;
  if (upload_acq_params(c, theta, m, prob, L, nullptr, 0)) return -1;
  if (run_predict(c, BOCF_ADD_NOISE | BOCF_CLIP, true, true)) return -1;
  AcqArgs a{};
  a.mean = c->mean.as<double>(); a.var = c->var.as<double>(); a.ld = c->pred_cap;
  a.m = m; a.C = c->C; a.L = L; a.kind = kind; a.util_kind = BOCF_UTIL_LINEAR; a.theta_dim = m;
  a.theta = c->theta.as<double>(); a.prob = c->prob.as<double>(); a.best = c->best.as<double>();
  a.util_params = c->params.as<double>(); a.acq = c->acq.as<double>();
  a.dmean = c->dmean.as<double>(); a.dvar = c->dvar.as<double>(); a.ldg = c->pred_cap; a.d = c->d; a.dacq = c->dacq.as<double>();
  if (acq_over_hyper_samples(c, a, m, 1, launch_acq_linear_grad)) return -1;
  return finish_acq_grad(c, acq_out, dacq_out);
}

extern "C" int bocf_set_mc_samples(bocf_ctx* c, const double* W, int S) {
  if (!c || !c->fitted || !W || S < 1) return fail("bocf_set_mc_samples", "model not fitted / bad samples");
  HIPCHK(hipSetDevice(c->device));
  const int m = group_size(c, "bocf_set_mc_samples");     // W is (S, outputs per hyper-sample)
  if (m < 0) return -1;
  std::vector<double> wt((size_t)m * S);
  for (int s = 0; s < S; ++s)
    for (int j = 0; j < m; ++j) wt[(long)j * S + s] = W[(long)s * m + j];
  if (c->Wt.ensure(sizeof(double) * wt.size())) return -1;
  HIPCHK(hipMemcpyAsync(c->Wt.p, wt.data(), sizeof(double) * wt.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->S_mc = S;
  return 0;
}

extern "C" int bocf_acq_mc(bocf_ctx* c, int kind, int util_kind, const double* util_params, int n_util_params, const double* theta,
                           int theta_dim, const double* prob, int L, double* acq_out) {
  if (!c || !c->fitted) return fail("bocf_acq_mc", "model not fitted");
  if (kind != BOCF_ACQ_EI && kind != BOCF_ACQ_PI) return fail("bocf_acq_mc", "unknown acquisition kind");
  if (util_kind < 0 || util_kind > BOCF_UTIL_ROSENBROCK) return fail("bocf_acq_mc", "unknown utility kind");
  if (c->S_mc < 1) return fail("bocf_acq_mc", "no Monte-Carlo samples set (bocf_set_mc_samples)");
  const int m = group_size(c, "bocf_acq_mc");
  if (m < 0) return -1;
  if ((util_kind == BOCF_UTIL_LINEAR || util_kind == BOCF_UTIL_NEG_SQ_DIST) && theta_dim != m) return fail("bocf_acq_mc", "theta_dim must equal m");
  if (util_kind == BOCF_UTIL_ROSENBROCK && (theta_dim < 1 || (m & 1))) return fail("bocf_acq_mc", "rosenbrock utility needs theta_dim >= 1 and even m");
  if (util_kind == BOCF_UTIL_NEG_EXP_COS && n_util_params != m) return fail("bocf_acq_mc", "neg_exp_cos needs m weights");
  if (n_util_params > 0 && !util_params) return fail("bocf_acq_mc", "util_params is null");
  HIPCHK(hipSetDevice(c->device));
  if (c->C == 0) return 0;
  if (upload_acq_params(c, theta, theta_dim, prob, L, util_params, n_util_params)) return -1;
  if (run_predict(c, BOCF_ADD_NOISE | BOCF_CLIP, true)) return -1;        // posterior_mean + posterior_variance (uEI_noiseless.py:73-74)
  AcqArgs a{};
  a.mean = c->mean.as<double>(); a.var = c->var.as<double>(); a.ld = c->pred_cap;
  a.m = m; a.C = c->C; a.L = L; a.kind = kind; a.util_kind = util_kind; a.theta_dim = theta_dim > 0 ? theta_dim : 1;
  a.theta = c->theta.as<double>(); a.prob = c->prob.as<double>(); a.best = c->best.as<double>();
  a.util_params = c->params.as<double>(); a.n_util_params = n_util_params;
  a.Wt = c->Wt.as<double>(); a.S = c->S_mc; a.acq = c->acq.as<double>();
  if (acq_over_hyper_samples(c, a, m, 0, launch_acq_mc)) return -1;
  return finish_acq(c, acq_out);
}

extern "C" int bocf_acq_mc_grad(bocf_ctx* c, int util_kind, const double* util_params, int n_util_params, const double* theta,
                                int theta_dim, const double* prob, int L, double* acq_out, double* dacq_out) {
  if (!c || !c->fitted) return fail("bocf_acq_mc_grad", "model not fitted");
  if (util_kind < 0 || util_kind > BOCF_UTIL_ROSENBROCK) return fail("bocf_acq_mc_grad", "unknown utility kind");
  if (c->S_mc < 1) return fail("bocf_acq_mc_grad", "no Monte-Carlo samples set (bocf_set_mc_samples)");
  const int m = group_size(c, "bocf_acq_mc_grad");
  if (m < 0) return -1;
  if ((util_kind == BOCF_UTIL_LINEAR || util_kind == BOCF_UTIL_NEG_SQ_DIST) && theta_dim != m) return fail("bocf_acq_mc_grad", "theta_dim must equal m");
  if (util_kind == BOCF_UTIL_ROSENBROCK && (theta_dim < 1 || (m & 1))) return fail("bocf_acq_mc_grad", "rosenbrock utility needs theta_dim >= 1 and even m");
  if (util_kind == BOCF_UTIL_NEG_EXP_COS && n_util_params != m) return fail("bocf_acq_mc_grad", "neg_exp_cos needs m weights");
  if (n_util_params > 0 && !util_params) return fail("bocf_acq_mc_grad", "util_params is null");
  if (c->d > 64) return fail("bocf_acq_mc_grad", "input dimension too large");
  HIPCHK(hipSetDevice(c->device));
  if (c->C == 0) return 0;
  if (upload_acq_params(c, theta, theta_dim, prob, L, util_params, n_util_params)) return -1;
  if (run_predict(c, BOCF_ADD_NOISE | BOCF_CLIP, true, true)) return -1;
  AcqArgs a{};
  a.mean = c->mean.as<double>(); a.var = c->var.as<double>(); a.ld = c->pred_cap;
  a.m = m; a.C = c->C; a.L = L; a.kind = BOCF_ACQ_EI; a.util_kind = util_kind; a.theta_dim = theta_dim > 0 ? theta_dim : 1;
  a.theta = c->theta.as<double>(); a.prob = c->prob.as<double>(); a.best = c->best.as<double>();
  a.util_params = c->params.as<double>(); a.n_util_params = n_util_params;
  a.Wt = c->Wt.as<double>(); a.S = c->S_mc; a.acq = c->acq.as<double>();
  a.dmean = c->dmean.as<double>(); a.dvar = c->dvar.as<double>(); a.ldg = c->pred_cap; a.d = c->d; a.dacq = c->dacq.as<double>();
  if (acq_over_hyper_samples(c, a, m, 0, launch_acq_mc_grad)) return -1;
  return finish_acq_grad(c, acq_out, dacq_out);
}

extern "C" int bocf_select_topk(bocf_ctx* c, int k, long long* idx_out, double* val_out) {
  if (!c || !c->have_acq) return fail("bocf_select_topk", "no acquisition vector on the device");
  if (k < 1 || k > 64 || !idx_out) return fail("bocf_select_topk", "k out of range (1..64) / null out");
  HIPCHK(hipSetDevice(c->device));
  const int nb = topk_num_blocks(c->C);
  if (c->blk_idx.ensure(sizeof(long long) * (size_t)nb * k) || c->blk_val.ensure(sizeof(double) * (size_t)nb * k) ||
      c->out_idx.ensure(8 * (size_t)128))                  // [0, k): indices, [k, 2k): values (k <= 64)
    return -1;
  {
    PhaseTimer t(c, "topk");
    launch_topk(c->acq.as<double>(), c->C, k, c->blk_idx.as<long long>(), c->blk_val.as<double>(), c->out_idx.as<long long>(),
                reinterpret_cast<double*>(c->out_idx.as<long long>() + k), c->stream);
  }
  // (indices and values sit next to each other in one allocation: ONE device-to-host copy)
  long long host[128];
  HIPCHK(hipMemcpyAsync(host, c->out_idx.p, 8 * (size_t)(val_out ? 2 * k : k), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(idx_out, host, sizeof(long long) * k);
  if (val_out) memcpy(val_out, host + k, sizeof(double) * k);
  LAUNCHCHK();
  return 0;
}

extern "C" int bocf_profile_read(bocf_ctx* c, double* ms_out, long long* launches_out, double* flops_out, int reset) {
  if (!c) return fail("bocf_profile_read", "null ctx");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  double ms = 0.0;
  for (auto& pr : c->events) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, pr.first, pr.second));
    ms += t;
  }
  if (ms_out) *ms_out = ms;
  if (launches_out) *launches_out = (long long)c->events.size();
  if (flops_out) *flops_out = c->prof_flops;
  if (reset) {
    drop_events(c);
    c->prof_flops = 0.0;
  }
  return 0;
}

extern "C" int bocf_get_stat(bocf_ctx* c, const char* name, long long* value_out) {
  if (!c || !name || !value_out) return fail("bocf_get_stat", "null argument");
  if (!strcmp(name, "sched_timeouts")) *value_out = c->sched_timeouts;
  else if (!strcmp(name, "gated_schedules_off")) *value_out = c->gated_off;
  else if (!strcmp(name, "last_schedule")) *value_out = c->last_schedule;
  else if (!strcmp(name, "early_inverse")) *value_out = c->early_inverse_started;
  else if (!strcmp(name, "cu_masks_ok")) *value_out = c->cu_masks_ok;
  else if (!strcmp(name, "comm_world")) *value_out = c->comm ? c->world : 0;
  else if (!strcmp(name, "kstar_workspace_bytes")) *value_out = (long long)c->Kstar.cap;
  else return fail("bocf_get_stat", "unknown statistic");
  return 0;
}

extern "C" int bocf_profile_phase(bocf_ctx* c, const char* name, double* ms_out, long long* count_out, int reset) {
  if (!c || !name) return fail("bocf_profile_phase", "null argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  double ms = 0.0;
  long long n = 0;
  auto it = c->phases.find(name);
  if (it != c->phases.end()) {
    for (auto& pr : it->second) {
      float t = 0.f;
      HIPCHK(hipEventElapsedTime(&t, pr.first, pr.second));
      ms += t;
      ++n;
    }
    if (reset) {
      for (auto& pr : it->second) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
      }
      c->phases.erase(it);
    }
  }
  if (ms_out) *ms_out = ms;
  if (count_out) *count_out = n;
  return 0;
}
