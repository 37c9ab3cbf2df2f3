// C ABI of libbocf_hip.so (declared in include/bocf_hip.h): context, memory, and the launch
// sequences of fit / predict / acquisition / selection.  No torch types, no CPU fallback.
#include "bocf_ctx.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_err;
int bocf_fail(const char* what, const char* detail) {
  g_err = std::string(what) + ": " + (detail ? detail : "");
  return -1;
}

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
  DevBuf* bufs[] = {&c->R32, &c->X, &c->Xs, &c->S, &c->R, &c->RT, &c->E, &c->ET, &c->T, &c->yc, &c->tvec, &c->alpha, &c->lml, &c->jit, &c->hypd,
                    &c->info, &c->mu_train, &c->rvec, &c->dvec, &c->hmc_buf, &c->Xc, &c->Kstar, &c->meanpart, &c->sumsq, &c->mean, &c->var, &c->acq, &c->Vbuf, &c->dmean, &c->dvar, &c->dacq, &c->Vs, &c->Ws, &c->theta,
                    &c->prob, &c->best, &c->params, &c->Wt, &c->blk_idx, &c->blk_val, &c->out_idx, &c->out_val, &c->gpart, &c->gout, &c->pack, &c->gidx,
                    &c->gval, &c->shard_meta, &c->chol_flags};
  for (DevBuf* b : bufs) b->release();
  if (c->infer_out) (void)hipHostFree(c->infer_out);
  for (hipEvent_t ev : c->ev_parts) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : c->ev_chol) (void)hipEventDestroy(ev);
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  for (hipStream_t st : {c->s_res, c->s_res2, c->s_hi, c->s_bulk, c->s_inv})
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
static bool opt_gemm_waves_ok(long long v) { return v == 4 || v == 8; }
static bool opt_lookahead_ok(long long v) { return v == -1 || v == 0 || v == 2 || v == 5; }
static bool opt_swizzle_ok(long long v) { return v == -1 || v == 0 || v == 1 || v == 2 || (v >= 100 && v < 164) || (v >= 256 && v <= 258); }
#ifdef BOCF_PROBES
static bool opt_potrf_ok(long long v) { return (v >= 0 && v <= 2) || (v >= 11 && v <= 14); }
#endif
static const OptDesc g_options[] = {
    {"chunk", 128, 1 << 24, 0, [](bocf_ctx* c, long long v) { c->chunk = (long)round_up((int)v, 128); }, nullptr, "candidates per pass (rounded up to 128)"},
    {"workspace_mb", 1, 1 << 20, 0, [](bocf_ctx* c, long long v) { c->workspace_mb = (long)v; }, nullptr, "cap of the per-pass K* workspace"},
    {"profile", 0, 1, 0, [](bocf_ctx* c, long long v) { c->profile = v != 0; }, nullptr, "HIP events around the dominant kernel and the named phases"},
    {"predict_f32", 0, 1, 1, [](bocf_ctx* c, long long v) { c->predict_f32 = v != 0; }, nullptr, "fp32 variance contraction (BASELINE configs[4])"},
    {"fused_infer", 0, 1, 0, [](bocf_ctx* c, long long v) { c->fused_infer = v != 0; }, nullptr, "one fused launch per inference for N <= 128"},
    {"reuse_data", 0, 1, 1, [](bocf_ctx* c, long long v) { c->reuse_data = v != 0; }, nullptr, "next fits reuse the resident X / Y"},
    {"skip_mu_train", 0, 1, 1, [](bocf_ctx* c, long long v) { c->skip_mu_train = v != 0; }, nullptr, "do not refresh the mean at the training inputs"},
    {"aggregate", 0, 8, 0, [](bocf_ctx* c, long long v) { c->aggregate = (int)v; }, nullptr, "panels per trailing update (0 = by size)"},
    {"lookahead", -1, 5, 0, [](bocf_ctx* c, long long v) { c->lookahead = (int)v; }, opt_lookahead_ok,
     "factorization schedule: -1 by size, 0 single stream, 2 reserved-CU chain, 5 persistent chain (experimental)"},
    {"lookahead_min_nb", 2, 1 << 20, 0, [](bocf_ctx* c, long long v) { c->lookahead_min_nb = (int)v; }, nullptr, "reserved-CU schedule from this many panels"},
    {"gemm_waves", 4, 8, 0, [](bocf_ctx* c, long long v) { c->gemm_waves = (int)v; }, opt_gemm_waves_ok, "waves per 128 x 128 tile of the store-epilogue GEMM (4 or 8)"},
    {"merge_x3", 0, 2, 0, [](bocf_ctx* c, long long v) { c->merge_x3 = (int)v; }, nullptr, "second product of an inverse merge in the three-buffer kernel"},
#ifdef BOCF_PROBES
    {"potrf_scalar", 0, 14, 2, [](bocf_ctx* c, long long v) { c->potrf_scalar = (int)v; }, opt_potrf_ok, "diagonal-block kernel: 0 / 1 / 2, 11..14 = TIMING-ONLY variants (wrong results)"},
#else
    {"potrf_scalar", 0, 2, 0, [](bocf_ctx* c, long long v) { c->potrf_scalar = (int)v; }, nullptr, "diagonal-block kernel: 0 factor wave / 1 scalar / 2 round-2a MFMA form"},
#endif
    {"shard_fit", 0, 1, 0, [](bocf_ctx* c, long long v) { c->shard_fit = v != 0; }, nullptr, "output-sharded fit over the communicator"},
    {"trsm_wave", 0, 1, 0, [](bocf_ctx* c, long long v) { c->trsm_wave = v != 0; }, nullptr, "row solves through the wave-level single-tile kernel"},
    {"overlap_inverse", -1, 1, 0, [](bocf_ctx* c, long long v) { c->overlap_inverse = (int)v; }, nullptr, "early part of the inverse underneath the factorization (-1 = by size)"},
    {"overlap", 0, 1, 0, [](bocf_ctx* c, long long v) { c->overlap = v != 0; }, nullptr, "K* build on a second stream"},
    {"small_path", 0, 1, 0, [](bocf_ctx* c, long long v) { c->small_path = v != 0; }, nullptr, "GEMV-shaped path for <= 16 candidates"},
    {"prefetch1", 0, 1, 0, [](bocf_ctx* c, long long v) { c->prefetch1 = v != 0; }, nullptr, "one-tile-deep staging in the 128-row variance kernel"},
    {"swizzle", -1, 258, 0, [](bocf_ctx* c, long long v) { c->swizzle = (int)v; }, opt_swizzle_ok, "variance-GEMM tiling: -1, 0, 1, 2, 100..163, 256, 257, 258"},
    {"hyper_samples", 1, 64, 1,
     [](bocf_ctx* c, long long v) {
       if ((int)v != c->hyper_samples) c->S_mc = 0;   // the transposed normals are laid out per group size
       c->hyper_samples = (int)v;
     },
     nullptr, "the fitted outputs are H hyper-samples x m / H model outputs"},
    {"acq_hyper_samples", 0, 64, 1, [](bocf_ctx* c, long long v) { c->acq_hyper_samples = (int)v; }, nullptr, "hyper-samples the acquisitions average over (0 = all)"},
    {"best_group", -1, 63, 1, [](bocf_ctx* c, long long v) { c->best_group = (int)v; }, nullptr, "whose best-so-far every hyper-sample uses (-1 = its own)"},
#ifdef BOCF_PROBES
    {"shard_fit_simulate", 0, 64, 2, [](bocf_ctx* c, long long v) { c->shard_fit_simulate = (int)v; }, nullptr, "TEST HOOK: one process plays all G ranks of a sharded fit"},
    {"kstar_valu_probe", 0, 4, 2, [](bocf_ctx* c, long long v) { c->kstar_valu_probe = (int)v; }, nullptr, "TIMING-ONLY variants of the two-buffer 256-row variance kernel (wrong results)"},
    {"test_diag_shift_1e12", -1000000000000LL, 1000000000000LL, 2, [](bocf_ctx* c, long long v) { c->test_diag_shift = (double)v * 1e-12; }, nullptr,
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

// ---------------------------------------------------------------------------------------------
// Cholesky (upper form, right-looking, NB = 128) of all m outputs at once.
//
// Per panel p the chain  diagonal block (one workgroup per output, ~41 us) -> row solve (one tile row) -> trailing
// update  is a dependency chain of short, latency-bound launches.  Schedules (option "lookahead"): 0 = everything on one stream, G panels
// per trailing update (option "aggregate"); 2 = the chain on reserved compute units with device-side counters (run_cholesky_reserved,
// the default for 12..24 panels); 5 = panel pairs with a persistent chain (run_cholesky_chain, experimental).
static GemmArgs trsm_args(bocf_ctx* c, int p, int W) {
  const int Np = c->Np;
  const long strideS = (long)Np * Np, strideE = (long)(Np / BOCF_TILE) * BOCF_TILE * BOCF_TILE;
  double* panel = c->S.as<double>() + (long)p * BOCF_TILE * Np + (long)(p + 1) * BOCF_TILE;
  GemmArgs g{};
  // U_p,> = E_p^T A_p,>   (in place)
  g.A = c->E.as<double>() + (long)p * BOCF_TILE * BOCF_TILE; g.lda = BOCF_TILE; g.strideA = strideE;
  g.B = panel; g.ldb = Np; g.strideB = strideS;
  g.Cin = nullptr; g.Cout = panel; g.ldc = Np; g.strideC = strideS;
  g.M = BOCF_TILE; g.Ncols = W; g.K = BOCF_TILE; g.kb = BOCF_TILE; g.alpha = 1.0; g.beta = 0.0;
  return g;
}

// the row solve of panel p over W columns: wave-level single-tile kernel (K = 128: latency, not throughput, decides) unless the
// option says otherwise
static void launch_trsm(bocf_ctx* c, int p, int W, hipStream_t st) {
  if (W <= 0) return;
  if (c->trsm_wave) {
    const int Np = c->Np;
    const long strideS = (long)Np * Np, strideE = (long)(Np / BOCF_TILE) * BOCF_TILE * BOCF_TILE;
    double* panel = c->S.as<double>() + (long)p * BOCF_TILE * Np + (long)(p + 1) * BOCF_TILE;
    launch_tile128(c->E.as<double>() + (long)p * BOCF_TILE * BOCF_TILE, BOCF_TILE, strideE, panel, Np, strideS, panel, Np, strideS, 1.0, 0.0, c->m, st,
                   W / BOCF_TILE);
  } else {
    launch_gemm_f64(trsm_args(c, p, W), c->m, 0, st);
  }
}

// A_>,> -= U_p,>^T U_p,> restricted to block rows [first, first + rows) of the trailing matrix (tiles on/above the diagonal)
static GemmArgs syrk_args(bocf_ctx* c, int p, int first, int rows, int W) {
  const int Np = c->Np;
  const long strideS = (long)Np * Np;
  double* panel = c->S.as<double>() + (long)p * BOCF_TILE * Np + (long)(p + 1) * BOCF_TILE;
  const long off = (long)first * BOCF_TILE;
  GemmArgs t{};
  t.A = panel + off; t.lda = Np; t.strideA = strideS;
  t.B = panel + off; t.ldb = Np; t.strideB = strideS;
  double* trail = c->S.as<double>() + ((long)(p + 1) * BOCF_TILE + off) * Np + (long)(p + 1) * BOCF_TILE + off;
  t.Cin = trail; t.Cout = trail; t.ldc = Np; t.strideC = strideS;
  t.M = rows * BOCF_TILE; t.Ncols = W - (int)off; t.K = BOCF_TILE; t.kb = BOCF_TILE; t.upper_only = 1; t.alpha = -1.0; t.beta = 1.0;
  return t;
}

// (Re)create the three masked streams for `want` reserved compute units.  Mask bit i selects CU (i / 8) of XCD (i % 8) on
// MI355X (tools/cumask_probe.hip), so 8 k reserved bits take k CUs from every XCD.
// Returns 0 = streams ready; 1 = the schedule does not apply (too few CUs for `want`, or the runtime refuses CU masks -- then
// cu_masks_ok is cleared) and the caller must fall through to a single-stream schedule; -1 = a HIP error (recorded).
static int ensure_reserved_streams(bocf_ctx* c, int want) {
  if (c->res_cus == want && c->s_res) return 0;
  for (hipStream_t* st : {&c->s_res, &c->s_res2, &c->s_hi, &c->s_bulk})
    if (*st) {
      (void)hipStreamDestroy(*st);
      *st = nullptr;
    }
  c->res_cus = 0;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, c->device));
  const int ncu = c->force_cu_count > 0 ? c->force_cu_count : prop.multiProcessorCount;
  if (want >= ncu / 2) return 1;                         // not applicable on this device / for this many outputs: no error, the caller falls through
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> res(words, 0u), rest(words, 0u);
  for (int i = 0; i < ncu; ++i) (i < want ? res : rest)[i / 32] |= 1u << (i % 32);
  int lo_prio = 0, hi_prio = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
  hipError_t e = hipExtStreamCreateWithCUMask(&c->s_res, (uint32_t)words, res.data());
  if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&c->s_res2, (uint32_t)words, res.data());
  if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&c->s_hi, (uint32_t)words, rest.data());
  if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&c->s_bulk, (uint32_t)words, rest.data());
  if (e != hipSuccess) {
    (void)hipGetLastError();
    for (hipStream_t* st : {&c->s_res, &c->s_res2, &c->s_hi, &c->s_bulk})
      if (*st) {
        (void)hipStreamDestroy(*st);
        *st = nullptr;
      }
    c->cu_masks_ok = 0;      // this runtime / box refuses CU masks: keep to the single-stream schedules
    return 1;
  }
  c->res_cus = want;
  return 0;
}

static void trtri_early(bocf_ctx* c, int h, hipStream_t st);
static int trtri_split(int nb);

// Right-looking blocked Cholesky whose serial chain runs alone on reserved compute units, with DEVICE-SIDE dependencies
// between its three streams (counters in memory, fit.hip: dep_signal / gate_kernel; stream events cost 10-25 us each here):
//
//   s_res  (reserved CUs)   potrf(p)  T1(p) S1(p)  potrf(p+1)  T1(p+1) S1(p+1)  potrf(p+2) ...
//   s_hi   (other CUs)              T2(p)   S2(p)          T2(p+1)   S2(p+1) ...
//   s_bulk (other CUs)                  bulkA(p) bulkB(p) ......... bulkA(p+1) bulkB(p+1) ...
//
//   potrf(p)  diagonal block p -> U_pp, E_p = U_pp^-1                    (one workgroup per output)        signals P(p)
//   T1(p)     U[p][p+1] = E_p^T A[p][p+1]                                 (ONE tile per output)             signals T1(p)
//   S1(p)     A[p+1][p+1] -= U[p][p+1]^T U[p][p+1]                        (all the next potrf needs)
//   T2(p)     U[p][c] = E_p^T A[p][c], c >= p+2                           (the rest of the row solve)       signals T2(p)
//   S2(p)     A[p+1][c] -= U[p][p+1]^T U[p][c], c >= p+2                  (the rest of block row p+1)       signals R(p)
//   bulkA(p)  block row p+2 of panel p's trailing update                  (then signal_kernel)             signals BA(p)
//   bulkB(p)  the rows below it
//
// Every tile receives its updates from different panels in whatever order the streams reach them (sums commute); what is
// enforced is mutual exclusion on a tile and completion before a tile is consumed -- by a gate in front of the consumer:
//   T1(p), S1(p): gate R(p-1), BA(p-1)      T2(p): gate P(p)      S2(p): gate T1(p), BA(p-1)      bulkA(p): gate T2(p)
// (bulkA(p-1) follows every older bulk on its in-order stream, so BA(p-1) stands for all of them.)  The chain per panel is
// potrf + two single-tile products + three kernel boundaries on CUs nobody else may use; a trailing update has two chain
// steps to finish before anything waits for it.
static int run_cholesky_reserved(bocf_ctx* c) {
  const int Np = c->Np, m = c->m, nb = Np / BOCF_TILE;
  const long strideS = (long)Np * Np, strideE = (long)nb * BOCF_TILE * BOCF_TILE;
  double* S = c->S.as<double>();
  while ((int)c->ev_chol.size() < 4) {
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c->ev_chol.push_back(ev);
  }
  const auto t_host0 = std::chrono::steady_clock::now();
  // counters: 5 per panel + the timeout word, in a block of their own (multiple of 16 bytes), zeroed before every schedule
  const size_t nflags = (size_t)((5 * nb + 1 + 3) / 4) * 4;
  if (c->chol_flags.ensure(sizeof(int) * nflags)) return -1;
  int* F = c->chol_flags.as<int>();
  HIPCHK(hipMemsetAsync(F, 0, sizeof(int) * nflags, c->stream));
  auto fP = [&](int p) { return F + 5 * p; };
  auto fT1 = [&](int p) { return F + 5 * p + 1; };
  auto fT2 = [&](int p) { return F + 5 * p + 2; };
  auto fR = [&](int p) { return F + 5 * p + 3; };
  auto fBA = [&](int p) { return F + 5 * p + 4; };
  int* ferr = F + 5 * nb;
  hipEvent_t ev0 = c->ev_chol[0], evE1 = c->ev_chol[1], evE2 = c->ev_chol[2], evE3 = c->ev_chol[3];
  HIPCHK(hipEventRecord(ev0, c->stream));
  for (hipStream_t st : {c->s_res, c->s_hi, c->s_bulk}) HIPCHK(hipStreamWaitEvent(st, ev0, 0));
  launch_potrf_diag(S, strideS, c->N, Np, 0, c->E.as<double>(), c->ET.as<double>(), strideE, c->info.as<int>(), m, c->s_res, fP(0));
  for (int p = 0; p + 1 < nb; ++p) {
    const int W = Np - (p + 1) * BOCF_TILE;                // trailing width after panel p (>= 128)
    const int nrest = W / BOCF_TILE - 1;                   // tiles right of column block p+1
    double* panel = S + (long)p * BOCF_TILE * Np + (long)(p + 1) * BOCF_TILE;              // U[p][p+1 ...]
    double* trail = S + (long)(p + 1) * BOCF_TILE * Np + (long)(p + 1) * BOCF_TILE;        // A[p+1][p+1 ...]
    const double* Ep = c->E.as<double>() + (long)p * BOCF_TILE * BOCF_TILE;
    const int prev_rest = nrest + 1;                       // nrest of panel p-1
    // ---- chain: T1(p), S1(p), potrf(p+1)
    if (p > 0) launch_gate(fR(p - 1), 4 * prev_rest * m, fBA(p - 1), prev_rest * m, ferr, c->s_res);
    launch_tile128(Ep, BOCF_TILE, strideE, panel, Np, strideS, panel, Np, strideS, 1.0, 0.0, m, c->s_res, 1, BOCF_TILE, fT1(p));     // T1(p)
    launch_tile128(panel, Np, strideS, panel, Np, strideS, trail, Np, strideS, -1.0, 1.0, m, c->s_res, 1, BOCF_TILE, nullptr);      // S1(p)
    launch_potrf_diag(S, strideS, c->N, Np, p + 1, c->E.as<double>(), c->ET.as<double>(), strideE, c->info.as<int>(), m, c->s_res, fP(p + 1));
    if (nrest <= 0) continue;                              // last panel pair: nothing right of column block p+1
    // ---- row work: T2(p), S2(p)
    launch_gate(fP(p), m, nullptr, 0, ferr, c->s_hi);
    launch_tile128(Ep, BOCF_TILE, strideE, panel + BOCF_TILE, Np, strideS, panel + BOCF_TILE, Np, strideS, 1.0, 0.0, m, c->s_hi, nrest, BOCF_TILE,
                   fT2(p));                                                                                                          // T2(p)
    launch_gate(fT1(p), 4 * m, p > 0 ? fBA(p - 1) : nullptr, prev_rest * m, ferr, c->s_hi);
    launch_tile128(panel, Np, strideS, panel + BOCF_TILE, Np, strideS, trail + BOCF_TILE, Np, strideS, -1.0, 1.0, m, c->s_hi, nrest, BOCF_TILE,
                   fR(p));                                                                                                           // S2(p)
    // ---- the part of the inverse that needs only block rows [0, h) of U starts as soon as row h-1 is solved, on its own stream
    //      (complement CUs): from here on the chain sets the pace and the chip is mostly idle
    {
      const bool want = c->overlap_inverse > 0 || (c->overlap_inverse < 0 && nb >= 16 && (c->sched_m > 0 ? c->sched_m : m) >= 2);
      if (want && c->s_inv && nb >= 8 && p == trtri_split(nb) - 1) {
        HIPCHK(hipStreamWaitEvent(c->s_inv, ev0, 0));
        launch_gate(fT2(p), 4 * nrest * m, fT1(p), 4 * m, ferr, c->s_inv);
        trtri_early(c, trtri_split(nb), c->s_inv);
        HIPCHK(hipEventRecord(c->ev_inv_early, c->s_inv));
        c->early_inverse_started = 1;
      }
    }
    // ---- trailing update below block row p+1
    launch_gate(fT2(p), 4 * nrest * m, nullptr, 0, ferr, c->s_bulk);
    launch_gemm_f64(syrk_args(c, p, 1, 1, W), m, 0, c->s_bulk);                                   // bulkA(p): block row p+2
    launch_signal(fBA(p), nrest * m, c->s_bulk);           // (the GEMM kernel is not instrumented: the kernel boundary is its release)
    if (nrest - 1 > 0) launch_gemm_f64(syrk_args(c, p, 2, nrest - 1, W), m, 0, c->s_bulk);        // bulkB(p)
  }
  HIPCHK(hipEventRecord(evE1, c->s_res));
  HIPCHK(hipEventRecord(evE2, c->s_hi));
  HIPCHK(hipEventRecord(evE3, c->s_bulk));
  for (hipEvent_t ev : {evE1, evE2, evE3}) HIPCHK(hipStreamWaitEvent(c->stream, ev, 0));
  c->chol_flags_used = 1;
  if (getenv("BOCF_DBG")) {
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "run_cholesky_reserved: host enqueue %.1f us for %d panels\n", std::chrono::duration<double, std::micro>(t1 - t_host0).count(), nb);
  }
  return 0;
}

static void trtri_early(bocf_ctx* c, int h, hipStream_t st);
static int trtri_split(int nb);

// Panel pairs with a PERSISTENT chain (option "lookahead" = 5): the aggregated pair schedule (one K = 256 trailing update per two
// panels) with lookahead -- the next pair's serial work underneath the bulk of this pair's trailing update -- and the chain's four
// kernels per pair replaced by two kernels that are launched ONCE and stay resident on the reserved compute units (fit.hip:
// chol_chain_potrf_kernel, chol_chain_tile_kernel), and everything else -- row products, trailing updates -- on ONE bulk stream behind
// single-wave gate kernels.  Why: in a plain run every kernel boundary of the chain that waited for another queue cost 17-30 us on this
// runtime (tools/dbg_timeline.py; 4.0 ms of Cholesky under rocprofv3 became 5.4-5.8 ms); here the chain has no kernel boundary at all
// and the only queue that dispatches work after the start is the bulk stream.
//
//   s_res / s_res2 (reserved CUs)   chain: [BA(g-1)] potrf(p0) -> T1 -> S1 -> potrf(p1)        per output, counters P0 T1 S1 P1
//   s_bulk         (other CUs)      [P0] T2(p0)  [T1] S2  [P1] T2'(p1)  bulkA(g) -> BA(g)  bulkB(g)
//
// T2: U[p0][c] = E_p0^T A[p0][c];  S2: A[p1][c] -= U[p0][p1]^T U[p0][c];  T2': U[p1][c] = E_p1^T A[p1][c]   (c >= p0 + 2)
// bulkA(g): block rows p0 + 2, p0 + 3 of  A[r][c] -= U[p0..p1][r]^T U[p0..p1][c]  (all the next pair's chain touches);  bulkB(g): the rows below.
// Same kernels on the same tiles in the same order per tile as the other pair schedules: the same factor bit for bit.
static int run_cholesky_chain(bocf_ctx* c) {
  const int Np = c->Np, m = c->m, nb = Np / BOCF_TILE, ng = nb / 2;
  const long strideS = (long)Np * Np, strideE = (long)nb * BOCF_TILE * BOCF_TILE;
  double* S = c->S.as<double>();
  while ((int)c->ev_chol.size() < 4) {
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c->ev_chol.push_back(ev);
  }
  const auto t_host0 = std::chrono::steady_clock::now();
  const int mpad = (m + 15) / 16 * 16;
  // counters: 4 per pair and output | BA per pair | RW per pair | (the time-out word where bocf_fit reads it: index 5 nb)
  const size_t nF = (size_t)4 * ng * mpad, total = (size_t)(5 * nb + 4) + nF + 2 * (size_t)ng + 8;
  if (c->chol_flags.ensure(sizeof(int) * total)) return -1;
  int* base = c->chol_flags.as<int>();
  HIPCHK(hipMemsetAsync(base, 0, sizeof(int) * total, c->stream));
  int* ferr = base + 5 * nb;
  int* F = base + 5 * nb + 4;
  int* BA = F + nF;
  int* RW = BA + ng;
  int* resident = RW + ng + 2;
  hipEvent_t ev0 = c->ev_chol[0], evE1 = c->ev_chol[1], evE2 = c->ev_chol[2], evE3 = c->ev_chol[3];
  HIPCHK(hipEventRecord(ev0, c->stream));
  for (hipStream_t st : {c->s_res, c->s_res2, c->s_bulk}) HIPCHK(hipStreamWaitEvent(st, ev0, 0));
  launch_chol_chain(S, strideS, Np, c->E.as<double>(), c->ET.as<double>(), strideE, c->info.as<int>(), F, mpad, BA, ferr, resident, m, c->s_res,
                    c->s_res2);
  const int h = trtri_split(nb);
  for (int g = 0; g < ng; ++g) {
    const int p0 = 2 * g, p1 = p0 + 1;
    const int W = Np - (p0 + 2) * BOCF_TILE;               // width of the trailing matrix behind the pair
    const int nrest = W / BOCF_TILE;                       // tiles right of column block p1
    if (nrest <= 0) break;
    double* row0 = S + (long)p0 * BOCF_TILE * Np + (long)p1 * BOCF_TILE;                  // U[p0][p1 ...]
    double* row1 = S + (long)p1 * BOCF_TILE * Np + (long)p1 * BOCF_TILE;                  // A[p1][p1 ...]
    const double* E0 = c->E.as<double>() + (long)p0 * BOCF_TILE * BOCF_TILE;
    const double* E1 = c->E.as<double>() + (long)p1 * BOCF_TILE * BOCF_TILE;
    launch_gate_multi(F + (4 * g + 0) * mpad, m, 1, ferr, c->s_bulk, 500000 + g * 10 + 0);
    launch_tile128(E0, BOCF_TILE, strideE, row0 + BOCF_TILE, Np, strideS, row0 + BOCF_TILE, Np, strideS, 1.0, 0.0, m, c->s_bulk, nrest, BOCF_TILE,
                   nullptr);                                                                                                         // T2
    launch_gate_multi(F + (4 * g + 1) * mpad, m, 4, ferr, c->s_bulk, 500000 + g * 10 + 1);
    launch_tile128(row0, Np, strideS, row0 + BOCF_TILE, Np, strideS, row1 + BOCF_TILE, Np, strideS, -1.0, 1.0, m, c->s_bulk, nrest, BOCF_TILE,
                   nullptr);                                                                                                         // S2
    launch_gate_multi(F + (4 * g + 3) * mpad, m, 1, ferr, c->s_bulk, 500000 + g * 10 + 3);
    launch_tile128(E1, BOCF_TILE, strideE, row1 + BOCF_TILE, Np, strideS, row1 + BOCF_TILE, Np, strideS, 1.0, 0.0, m, c->s_bulk, nrest, BOCF_TILE,
                   RW + g);                                                                                                          // T2'
    // ---- the part of the inverse that needs only block rows [0, h) of U, as soon as they are final
    {
      const bool want = c->overlap_inverse > 0 || (c->overlap_inverse < 0 && nb >= 16 && (c->sched_m > 0 ? c->sched_m : m) >= 2);
      if (want && c->s_inv && nb >= 8 && !c->early_inverse_started && p1 >= h - 1) {
        HIPCHK(hipStreamWaitEvent(c->s_inv, ev0, 0));
        launch_gate(RW + g, 4 * nrest * m, nullptr, 0, ferr, c->s_inv);
        trtri_early(c, h, c->s_inv);
        HIPCHK(hipEventRecord(c->ev_inv_early, c->s_inv));
        c->early_inverse_started = 1;
      }
    }
    auto bulk = [&](int first, int rows) {
      GemmArgs t{};
      const long off = (long)first * BOCF_TILE;
      double* urows = S + (long)p0 * BOCF_TILE * Np + (long)(p0 + 2) * BOCF_TILE + off;
      t.A = urows; t.lda = Np; t.strideA = strideS;
      t.B = urows; t.ldb = Np; t.strideB = strideS;
      double* trail = S + ((long)(p0 + 2) * BOCF_TILE + off) * Np + (long)(p0 + 2) * BOCF_TILE + off;
      t.Cin = trail; t.Cout = trail; t.ldc = Np; t.strideC = strideS;
      t.M = rows * BOCF_TILE; t.Ncols = W - (int)off; t.K = 2 * BOCF_TILE; t.kb = 2 * BOCF_TILE; t.upper_only = 1; t.alpha = -1.0; t.beta = 1.0;
      launch_gemm_f64(t, m, 0, c->s_bulk);
    };
    bulk(0, nrest < 2 ? nrest : 2);                        // bulkA(g)
    launch_signal(BA + g, 1, c->s_bulk);                   // (the kernel boundary behind the GEMM is its release)
    if (nrest > 2) bulk(2, nrest - 2);                     // bulkB(g)
  }
  HIPCHK(hipEventRecord(evE1, c->s_res));
  HIPCHK(hipEventRecord(evE2, c->s_res2));
  HIPCHK(hipEventRecord(evE3, c->s_bulk));
  for (hipEvent_t ev : {evE1, evE2, evE3}) HIPCHK(hipStreamWaitEvent(c->stream, ev, 0));
  c->chol_flags_used = 1;
  if (getenv("BOCF_DBG_FLAGS"))
    fprintf(stderr, "run_cholesky_chain: host enqueue %.1f us for %d pairs\n",
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count(), ng);
  return 0;
}

// Called by the single-stream Cholesky schedules right after the row solve of panel p: once block rows [0, h) of U are final
// the part of the inverse that needs nothing else starts on the second stream, underneath the rest of the factorization
// (whose second half is a chain of short launches that leaves most of the chip idle).
static int maybe_start_early_inverse(bocf_ctx* c, int p) {
  const int nb = c->Np / BOCF_TILE;
  const bool want = c->overlap_inverse > 0 || (c->overlap_inverse < 0 && nb >= 32 && (c->sched_m > 0 ? c->sched_m : c->m) >= 2);
  if (!want || nb < 8 || c->early_inverse_started || !c->s_inv) return 0;
  if (p != trtri_split(nb) - 1) return 0;
  HIPCHK(hipEventRecord(c->ev_half, c->stream));
  HIPCHK(hipStreamWaitEvent(c->s_inv, c->ev_half, 0));
  trtri_early(c, trtri_split(nb), c->s_inv);
  HIPCHK(hipEventRecord(c->ev_inv_early, c->s_inv));
  c->early_inverse_started = 1;
  return 0;
}

static int run_cholesky_impl(bocf_ctx* c);
static int run_cholesky(bocf_ctx* c) {
  const char* tl = getenv("BOCF_DBG_TL");
  if (tl) {
    HIPCHK(hipStreamSynchronize(c->stream));
    dbg_tl_start();
  }
  const int rc = run_cholesky_impl(c);
  if (tl && rc == 0) dbg_tl_dump(tl);
  return rc;
}
static int run_cholesky_impl(bocf_ctx* c) {
  const int Np = c->Np, m = c->m, nb = Np / BOCF_TILE;
  const long strideS = (long)Np * Np, strideE = (long)nb * BOCF_TILE * BOCF_TILE;
  double* S = c->S.as<double>();
  c->early_inverse_started = 0;
  set_potrf_scalar(c->potrf_scalar);                     // (the kernel choice is a launcher-level switch; contexts are not thread-safe)
  set_gemm_store_waves(c->gemm_waves);
  // schedule: option "lookahead" = 2 (default by size: nb >= 8, at most 64 factorizations) -> reserved-CU lookahead
  // reserved-CU schedule with device-side dependencies: where the CHAIN of diagonal blocks sets the pace (few panels, or few
  // outputs per panel) it wins -- N = 2048 m = 4: 2.83 -> 2.52 ms, N = 3072: 5.4 -> 4.6, N = 4096 m = 1: 5.83 -> 4.57 -- where the
  // trailing updates do (N >= 6144 with m = 4: 17.7 vs 18.9 ms) the aggregated single-stream schedule below does.
  // "lookahead" = 2 forces it, -1 (default) chooses by size, 0 never uses it.  (Removed in round 3, all measured slower in plain runs and
  // kept until then for A/B: 1 = next panel's diagonal block + row solve on a second stream with stream events, 3 / 4 = panel pairs with
  // lookahead on two / three masked streams; their numbers are in DESIGN.md 10 and profiles/r02.)
  const int m_sched = c->sched_m > 0 ? c->sched_m : m;     // (a shard helper chooses as the replicated fit of ALL outputs would)
  const bool reserved_auto = c->lookahead < 0 && nb >= 12 && (nb <= 24 || (nb <= 32 && m_sched <= 2));
  // The gated (multi-stream) schedules are not used: after dependency time-outs (gated_off), for the redo of an attempt that timed out
  // (sched_retry), and for the FIRST factorization of a context -- it pays the one-time costs (code-object loads, allocations, stream
  // creation) that would otherwise sit between the launch of a polling kernel and the launch of the kernel it waits for.
  const bool gated_ok = c->cu_masks_ok && !c->gated_off && !c->sched_retry && c->fits_done > 0;
  c->sched_retry = 0;
  if (c->lookahead == 5 && gated_ok && nb >= 4 && nb % 2 == 0 && m <= 64) {       // experimental: measured slower
    const int rs = ensure_reserved_streams(c, 8 * chol_chain_cus_per_xcd(m));
    if (rs < 0) return -1;
    if (rs == 0) {
      c->last_schedule = 5;
      return run_cholesky_chain(c);
    }
  }
  if ((c->lookahead == 2 || reserved_auto) && gated_ok && nb >= (c->lookahead == 2 ? 2 : c->lookahead_min_nb) && m <= 64 && c->aggregate <= 0) {
    const int rs = ensure_reserved_streams(c, ((m + 7) / 8) * 8);
    if (rs < 0) return -1;
    if (rs == 0) {
      c->last_schedule = 2;
      return run_cholesky_reserved(c);
    }
  }
  c->last_schedule = 0;
  // measured (m = 4): N=2048 4 % slower, N=4096 3 % faster, N=8192 5 % faster -- the diagonal-block workgroup runs 1.6-2x
  // slower when it shares its CU with trailing-update waves, which eats most of what the overlap hides
  // measured (m = 4, ms): N=2048 3.82 / 3.90 / 4.13 for G = 1 / 2 / 4; N=4096 11.45 / 11.17 / 11.45; N=8192 56.3 / 50.4 / 48.7
  // re-measured with the MFMA diagonal-block kernel and the row-staged epilogue (profiles/r02/fit_schedule_sweep.txt):
  // G = 1 is best up to N = 3072, 2 at 4096, 3 at 6144 and 8192
  // (G = 3 at N = 4096 is 0.15 ms faster than G = 2 with the factor-wave diagonal kernel, but at cond(Ky) ~ 4e9 the other summation order moves
  // two of config 3's small acquisition values by 2.5e-5 relative, past the 1e-5 gate of test_config3_full_size: not taken)
  const int G_auto = nb >= 48 ? 3 : (nb >= 32 ? 2 : 1);
  const int G_use = c->aggregate > 0 ? c->aggregate : G_auto;
  if (G_use > 1 && nb >= 2 * G_use) {
    // G panels per trailing update: the trailing matrix is read-modify-written once per G panels (its HBM traffic, not
    // flops, is what the K = 128 updates cost); inside a group each new block row first receives the group's finished
    // rows as ONE thin update with K = 128 * (rows so far).
    const int G = G_use;
    for (int p0 = 0; p0 < nb; p0 += G) {
      const int g = (nb - p0) < G ? (nb - p0) : G;
      for (int q = 0; q < g; ++q) {
        const int p = p0 + q;
        const int W = Np - (p + 1) * BOCF_TILE;
        if (q > 0) {
          // block row p -= U_{p0..p-1, p}^T U_{p0..p-1, p..}   (K = 128 q)
          GemmArgs t{};
          double* rows = S + (long)p0 * BOCF_TILE * Np + (long)p * BOCF_TILE;
          t.A = rows; t.lda = Np; t.strideA = strideS;
          t.B = rows; t.ldb = Np; t.strideB = strideS;
          double* row = S + (long)p * BOCF_TILE * Np + (long)p * BOCF_TILE;
          t.Cin = row; t.Cout = row; t.ldc = Np; t.strideC = strideS;
          t.M = BOCF_TILE; t.Ncols = W + BOCF_TILE; t.K = q * BOCF_TILE; t.kb = q * BOCF_TILE; t.alpha = -1.0; t.beta = 1.0;
          if (c->trsm_wave)   // one block row, short K: the wave-level kernel (latency-bound either way, half the time)
            launch_tile128(rows, Np, strideS, rows, Np, strideS, row, Np, strideS, -1.0, 1.0, m, c->stream, (W + BOCF_TILE) / BOCF_TILE, q * BOCF_TILE);
          else
            launch_gemm_f64(t, m, 0, c->stream);
        }
        launch_potrf_diag(S, strideS, c->N, Np, p, c->E.as<double>(), c->ET.as<double>(), strideE, c->info.as<int>(), m, c->stream);
        launch_trsm(c, p, W, c->stream);
        if (maybe_start_early_inverse(c, p)) return -1;
      }
      const int pe = p0 + g;                            // first block row after the group
      const int W = Np - pe * BOCF_TILE;
      if (W > 0) {
        GemmArgs t{};
        double* rows = S + (long)p0 * BOCF_TILE * Np + (long)pe * BOCF_TILE;
        t.A = rows; t.lda = Np; t.strideA = strideS;
        t.B = rows; t.ldb = Np; t.strideB = strideS;
        double* trail = S + (long)pe * BOCF_TILE * Np + (long)pe * BOCF_TILE;
        t.Cin = trail; t.Cout = trail; t.ldc = Np; t.strideC = strideS;
        t.M = W; t.Ncols = W; t.K = g * BOCF_TILE; t.kb = g * BOCF_TILE; t.upper_only = 1; t.alpha = -1.0; t.beta = 1.0;
        launch_gemm_f64(t, m, 0, c->stream);
      }
    }
    return 0;
  }
  for (int p = 0; p < nb; ++p) {
    launch_potrf_diag(S, strideS, c->N, Np, p, c->E.as<double>(), c->ET.as<double>(), strideE, c->info.as<int>(), m, c->stream);
    const int W = Np - (p + 1) * BOCF_TILE;
    if (W <= 0) break;
    launch_trsm(c, p, W, c->stream);
    if (maybe_start_early_inverse(c, p)) return -1;
    launch_gemm_f64(syrk_args(c, p, 0, W / BOCF_TILE, W), m, 0, c->stream);
  }
  return 0;
}

// R = U^-1 (upper) by recursive doubling over the 128-blocks: the diagonal tiles are the E_p of the
// diagonal-block kernel; two neighbouring inverted blocks [lo,mid), [mid,hi) merge with
//     R12 = -(R11 * U12) * R22
// as two GEMMs.  All merges of one level are independent and run as ONE batched launch, so the whole inverse is
// ~log2(nb) levels of large GEMMs instead of nb dependent thin ones.  RT holds R^T (lower): the first product needs R11
// k-major.  The association (R11 U12) first matters twice: U12 enters as rows of the upper factor (no mirrored copy of U is
// needed), and the first product of a merge depends only on the LEFT half -- so everything that involves only the first h
// block rows (all their merges and the first product of the top-level merge) can run while the Cholesky is still busy
// with the block rows below (run_cholesky starts it on a second stream as soon as panel h-1 is solved).
//   first :  T'[r][c']   = sum_{kk >= r} R11[r][kk] U12[kk][c']        A = RT11, B = rows of U; then T'^T by a transpose
//   second:  RT21[c][r]  = -sum_{kk <= c} R22[kk][c] T'^T[kk][r]        A = rows of R22, B = T'^T; then R12 by a transpose
enum { MERGE_FIRST = 1, MERGE_SECOND = 2 };
static void merge_level(bocf_ctx* c, int lo, int w, int w2, int count, int which, hipStream_t st) {
  const int Np = c->Np, m = c->m;
  const long strideS = (long)Np * Np;
  const long dstep = (long)2 * w * BOCF_TILE * (Np + 1);            // next pair along the diagonal
  const long oLo = (long)lo * BOCF_TILE, oMid = (long)(lo + w) * BOCF_TILE;
  const int b1 = w * BOCF_TILE, b2 = w2 * BOCF_TILE;
  double* S = c->S.as<double>();
  double* R = c->R.as<double>();
  double* RT = c->RT.as<double>();
  double* T = c->T.as<double>();
  // Both products are arranged so that the contraction length depends on the ROW tile (whole rows of equal-length
  // workgroups, heaviest rows first): measured 0.85 ms against 1.03-1.09 ms for the same product with the length varying
  // along a row (top level of N = 4096).  The price is one extra transpose per level.
  if (which & MERGE_FIRST) {
    // T'[r][c'] = sum_{kk >= r} R11[r][kk] U12[kk][c']      A = RT11 (k-major R11), B = rows of U;  into T at (lo, mid)
    GemmArgs g{};
    g.A = RT + oLo * Np + oLo; g.lda = Np; g.strideA = strideS; g.strideA2 = dstep;
    g.B = S + oLo * Np + oMid; g.ldb = Np; g.strideB = strideS; g.strideB2 = dstep;
    g.Cin = nullptr; g.Cout = T + oLo * Np + oMid; g.ldc = Np; g.strideC = strideS; g.strideC2 = dstep;
    g.M = b1; g.Ncols = b2; g.K = b1; g.kb = b1; g.kbeg_rt = BOCF_TILE; g.alpha = 1.0; g.batch1 = m;
    g.swizzle = 2;      // row-tile-major across the whole batch (all outputs' heaviest row tiles first): inverse 2.55 -> 2.03 ms at config 3
    launch_gemm_f64(g, m * count, 0, st);
    // T'^T into T at (mid, lo): the k-major operand of the second product
    launch_transpose_block(T, T, strideS, Np, (int)oLo, (int)oMid, b1, b2, count, 2 * w * BOCF_TILE, m, st);
  }
  if (which & MERGE_SECOND) {
    // RT21[c][r] = R12[r][c] = -sum_{kk <= c} R22[kk][c] T'^T[kk][r]      A = rows of R22, B = T'^T;  straight into R^T
    GemmArgs h{};
    h.A = R + oMid * Np + oMid; h.lda = Np; h.strideA = strideS; h.strideA2 = dstep;
    h.B = T + oMid * Np + oLo; h.ldb = Np; h.strideB = strideS; h.strideB2 = dstep;
    h.Cin = nullptr; h.Cout = RT + oMid * Np + oLo; h.ldc = Np; h.strideC = strideS; h.strideC2 = dstep;
    h.M = b2; h.Ncols = b1; h.K = b2; h.kb = BOCF_TILE; h.krt = BOCF_TILE; h.rt_desc = 1; h.alpha = -1.0; h.batch1 = m;
    // the three-buffer triangular kernel with its store epilogue (the product has the variance's shape) from 4096 rows: measured inverse 6.80 -> 6.53 ms
    // at N = 8192, but 1.47 -> 1.55 at N = 4096 (2048-row products: 512 workgroups of very unequal length on 256 CUs suit the smaller tiles better)
    h.no_x3 = c->merge_x3 <= 0 || (c->merge_x3 == 1 && b2 < 4096);
    h.swizzle = 2;
    launch_gemm_f64(h, m * count, 0, st);
    // R12 = RT21^T
    launch_transpose_block(RT, R, strideS, Np, (int)oMid, (int)oLo, b2, b1, count, 2 * w * BOCF_TILE, m, st);
  }
}

static void copy_diag_range(bocf_ctx* c, int blk_lo, int blk_hi, hipStream_t st) {
  const int Np = c->Np, m = c->m, nb = Np / BOCF_TILE;
  const long strideS = (long)Np * Np, strideE = (long)nb * BOCF_TILE * BOCF_TILE;
  launch_copy_diag_blocks(c->E.as<double>(), strideE, c->R.as<double>(), strideS, Np, blk_lo, blk_hi, m, st);
  launch_copy_diag_blocks(c->ET.as<double>(), strideE, c->RT.as<double>(), strideS, Np, blk_lo, blk_hi, m, st);
}

// split of the inverse: h = the largest power of two below nb; blocks [0, h) form complete pairs at every level below h
static int trtri_split(int nb) {
  int h = 1;
  while (2 * h < nb) h *= 2;
  return h;
}

// everything of the inverse that needs only block rows [0, h) of U: runs on `st` as soon as those rows are final
static void trtri_early(bocf_ctx* c, int h, hipStream_t st) {
  const int nb = c->Np / BOCF_TILE;
  copy_diag_range(c, 0, h, st);
  for (int w = 1; w < h; w *= 2) merge_level(c, 0, w, w, h / (2 * w), MERGE_FIRST | MERGE_SECOND, st);
  merge_level(c, 0, h, nb - h, 1, MERGE_FIRST, st);
}

// the rest: the merges among block rows [h, nb) and the second product of the top-level merge
static void trtri_late(bocf_ctx* c, int h, hipStream_t st) {
  const int nb = c->Np / BOCF_TILE;
  copy_diag_range(c, h, nb, st);
  for (int w = 1; w < h; w *= 2) {
    const int full = nb / (2 * w);                       // pairs with two complete halves
    const int first = h / (2 * w);                       // pairs that lie inside [0, h): done early
    if (full > first) merge_level(c, h, w, w, full - first, MERGE_FIRST | MERGE_SECOND, st);
    const int g = full * 2 * w;                          // a trailing incomplete pair, if any
    if (g + w < nb && g >= h) merge_level(c, g, w, nb - (g + w), 1, MERGE_FIRST | MERGE_SECOND, st);
  }
  merge_level(c, 0, h, nb - h, 1, MERGE_SECOND, st);
}

static int run_trtri(bocf_ctx* c, bool early_done) {
  const int nb = c->Np / BOCF_TILE;
  if (early_done) {                                      // the first h block rows were inverted underneath the factorization
    trtri_late(c, trtri_split(nb), c->stream);
    return 0;
  }
  // everything here: every level is ONE batched launch over all its pairs (the early / late split would double the
  // launch count, which is what the small sizes are made of)
  copy_diag_range(c, 0, nb, c->stream);
  for (int w = 1; w < nb; w *= 2) {
    const int full = nb / (2 * w);                       // pairs with two complete halves
    if (full > 0) merge_level(c, 0, w, w, full, MERGE_FIRST | MERGE_SECOND, c->stream);
    const int g = full * 2 * w;                          // a trailing incomplete pair, if any
    if (g + w < nb) merge_level(c, g, w, nb - (g + w), 1, MERGE_FIRST | MERGE_SECOND, c->stream);
  }
  return 0;
}

static int nsplit_for(int Np, int Cpad, int m) {
  const int blocks = ((Cpad + 511) / 512) * m;        // cross_kernel: 256 threads x 2 columns per workgroup
  int ns = 2048 / (blocks > 0 ? blocks : 1);
  if (ns < 1) ns = 1;
  const int maxs = Np / BOCF_TILE;
  if (ns > maxs) ns = maxs;
  return ns;
}

// alpha = Ky^-1 yc = R (R^T yc) (exact_gaussian_inference.py:51), the log-marginal (:53) and -- unless the caller is an
// inference of a hyper-parameter update, which reads neither -- ONE step of iterative refinement with the residual yc - Ky alpha
// carried in double-double, and the posterior mean at the training inputs (multi_outputGP.py:176-180) as yc + ymean - dg alpha.
// Why: at BASELINE configs[2] (cond(Ky) ~ 4e9) any fp64 solve -- LAPACK's dpotrs as much as R (R^T yc) -- leaves ~4e-8 relative
// in alpha, i.e. ~1e-7 absolute in a posterior mean of size 1, and WHICH 1e-7 depends on the summation order of the factorization
// (panels per trailing update, ...).  Measured against oracle/truth_ld.c (long double end to end), tests/test_gpu_round3.py: one
// refinement step takes alpha to the floor set by the fp64 rounding of K itself (2e-9 relative) whatever schedule produced R, which
// is what makes the choice of schedule a matter of speed only.  Cost: one pass over K (rebuilt on the fly, N^2 m kernel values) and
// two more GEMVs per fit; the pass over K that the train mean used to take is gone.
static int solve_alpha(bocf_ctx* c, bool refine_and_train_mean) {
  const int N = c->N, Np = c->Np, m = c->m, nb = Np / BOCF_TILE;
  const long strideS = (long)Np * Np;
  launch_gemv_small_t(c->R.as<double>(), strideS, Np, c->yc.as<double>(), 1, Np, c->tvec.as<double>(), 1, m, c->stream);
  launch_gemv_upper_n(c->R.as<double>(), strideS, Np, c->tvec.as<double>(), c->alpha.as<double>(), m, c->stream);
  if (refine_and_train_mean) {
    if (c->meanpart.ensure(sizeof(double) * (size_t)2 * m * nb * Np) || c->rvec.ensure(sizeof(double) * (size_t)m * Np) ||
        c->dvec.ensure(sizeof(double) * (size_t)m * Np) || c->mu_train.ensure(sizeof(double) * (size_t)m * Np))
      return -1;
    launch_kalpha_dd(c->Xs.as<double>(), c->xs_stride, N, Np, c->d, c->kernel_id, c->hypd.as<KernHyp>(), c->jit.as<double>(), c->alpha.as<double>(),
                     c->meanpart.as<double>(), m, c->stream);
    launch_refine_rhs(c->meanpart.as<double>(), N, Np, c->yc.as<double>(), c->rvec.as<double>(), m, c->stream);
    launch_gemv_small_t(c->R.as<double>(), strideS, Np, c->rvec.as<double>(), 1, Np, c->tvec.as<double>(), 1, m, c->stream);
    launch_gemv_upper_n(c->R.as<double>(), strideS, Np, c->tvec.as<double>(), c->dvec.as<double>(), m, c->stream);
    launch_refine_apply(c->dvec.as<double>(), N, Np, c->hypd.as<KernHyp>(), c->jit.as<double>(), c->yc.as<double>(), c->alpha.as<double>(),
                        c->mu_train.as<double>(), N, m, c->stream);
  }
  launch_lml(c->S.as<double>(), strideS, N, Np, c->alpha.as<double>(), c->yc.as<double>(), c->lml.as<double>(), m, c->stream);
  return 0;
}

// X, the centred targets and the hyper-parameters onto the device (X, yc, hypd must be allocated).  With option
// "reuse_data" only the hyper-parameters move: X and Y are those of the previous call (same N, d, m).
static int stage_data(bocf_ctx* c, const double* X, const double* Y, int N, int Np, int d, int m, const double* variance,
                      const double* lengthscale, const double* noise) {
  const bool reuse = c->reuse_data && c->data_N == N && c->data_d == d && c->data_m == m && (int)c->hyp.size() == m;
  if (c->reuse_data && !reuse) return fail("bocf_fit / bocf_infer", "option reuse_data is set but N, d or m differ from the previous fit");
  if (reuse) {
    // same X and targets as the previous fit (HMC / optimiser inferences): only the hyper-parameters are uploaded
    for (int j = 0; j < m; ++j) {
      KernHyp& h = c->hyp[j];
      h.variance = variance[j]; h.noise = noise[j]; h.jitter = -c->test_diag_shift;
      for (int q = 0; q < BOCF_MAX_D; ++q) h.ls[q] = q < d ? lengthscale[(long)j * d + q] : 1.0;
    }
    HIPCHK(hipMemcpyAsync(c->hypd.p, c->hyp.data(), sizeof(KernHyp) * m, hipMemcpyHostToDevice, c->stream));   // c->hyp outlives the copy
  } else {
    // Standardize: subtract the mean only (normalizer.py:57-70)
    c->hyp.assign(m, KernHyp());
    std::vector<double> yc((size_t)m * Np, 0.0);
    for (int j = 0; j < m; ++j) {
      double s = 0.0;
      for (int i = 0; i < N; ++i) s += Y[(long)j * N + i];
      const double mean = s / N;
      KernHyp& h = c->hyp[j];
      h.variance = variance[j]; h.noise = noise[j]; h.ymean = mean; h.jitter = -c->test_diag_shift;
      for (int q = 0; q < BOCF_MAX_D; ++q) h.ls[q] = q < d ? lengthscale[(long)j * d + q] : 1.0;
      for (int i = 0; i < N; ++i) yc[(long)j * Np + i] = Y[(long)j * N + i] - mean;
    }
    HIPCHK(hipMemcpyAsync(c->X.p, X, sizeof(double) * N * d, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->hypd.p, c->hyp.data(), sizeof(KernHyp) * m, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->yc.p, yc.data(), sizeof(double) * (size_t)m * Np, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));   // host staging buffers go out of scope below
    c->data_N = N; c->data_d = d; c->data_m = m;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Output-sharded fit (SURVEY 8e "better"): the m factorizations are independent (multi_outputGP.py:64-95 builds one GPModel
// per output, :97-102 updates them one after the other), so rank r of G factorizes only the outputs of its contiguous
// share [j0, j1) -- with the ordinary bocf_fit, in a helper context on the same GPU -- and the ranks then exchange what
// PREDICTION needs: the inverse factor R_j (broadcast from its owner over xGMI, the m broadcasts in one RCCL group) and the
// small per-output vectors alpha_j, mean at the training inputs, log-marginal, jitter, status (every element has exactly one
// owner, the others hold zeros: ONE all-reduce(SUM)).  R^T (the k-major operand of the gradient path) is rebuilt locally by
// a transpose.  The upper factor U itself is NOT exchanged: entry points that need it (bocf_get_factor, bocf_append,
// bocf_lml_gradients) report that on a sharded fit and the caller refits unsharded.
// Bytes per rank: receives (m - m_local) Np^2 x 8 B (134 MB per output at N = 4096), sends m_local x that to every peer.
static void shard_range(int m, int G, int r, int* j0, int* j1) {
  const int base = m / G, rem = m % G;
  *j0 = r * base + (r < rem ? r : rem);
  *j1 = *j0 + base + (r < rem ? 1 : 0);
}

// The local share of a sharded fit: this rank's outputs through the ordinary bocf_fit of the helper context, results copied into
// R and the meta block.  Any failure comes back as -1 (error text recorded) WITHOUT returning from fit_sharded: the caller must still
// take part in the collectives, or every peer would wait for this rank forever.
static int fit_sharded_local(bocf_ctx* c, bocf_ctx* hctx, int G, int me, int simulate, const double* X, const double* Y, int N, int d, int m,
                             int kernel_id, const double* variance, const double* lengthscale, const double* noise, int max_jitter_tries,
                             size_t meta_w, std::vector<double>& meta_host) {
  const int Np = c->Np;
  const long strideS = (long)Np * Np;
  for (int r = 0; r < G; ++r) {
    if (!simulate && r != me) continue;
    int j0, j1;
    shard_range(m, G, r, &j0, &j1);
    const int ml = j1 - j0;
    if (ml <= 0) continue;
    std::vector<double> jit(ml, 0.0), lml(ml, 0.0);
    // INVARIANT: bocf_fit on the helper is synchronous (its stream is idle on return) and the copies below run on c->stream; with
    // more than one share per process (the simulate hook) c->stream is drained before the helper refits, because that refit
    // rewrites the buffers the copies read.
    const int rc = bocf_fit(hctx, X, Y + (size_t)j0 * N, N, d, ml, kernel_id, variance + j0, lengthscale + (size_t)j0 * d, noise + j0,
                            max_jitter_tries, jit.data(), lml.data());
    if (rc < 0) return -1;
    HIPCHK(hipSetDevice(c->device));
    std::vector<int> info(ml, 0);
    if (bocf_last_fit_info(hctx, info.data(), ml)) return -1;
    for (int j = 0; j < ml; ++j) {
      double* row = c->shard_meta.as<double>() + (size_t)(j0 + j) * meta_w;
      if (rc == 0) {
        if (r == me) {
          HIPCHK(hipMemcpyAsync(c->R.as<double>() + (size_t)(j0 + j) * strideS, hctx->R.as<double>() + (size_t)j * strideS, sizeof(double) * strideS,
                                hipMemcpyDeviceToDevice, c->stream));
        } else {     // (simulate hook only) a foreign share arrives the way it would over RCCL: upper tiles packed, then unpacked
          const size_t packed = (size_t)(Np / BOCF_TILE) * (Np / BOCF_TILE + 1) / 2 * BOCF_TILE * BOCF_TILE;
          launch_pack_upper_tiles(hctx->R.as<double>() + (size_t)j * strideS, Np, c->T.as<double>() + (size_t)(j0 + j) * packed, c->stream);
          launch_unpack_upper_tiles(c->T.as<double>() + (size_t)(j0 + j) * packed, Np, c->R.as<double>() + (size_t)(j0 + j) * strideS, c->stream);
        }
        HIPCHK(hipMemcpyAsync(row, hctx->alpha.as<double>() + (size_t)j * Np, sizeof(double) * Np, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(row + Np, hctx->mu_train.as<double>() + (size_t)j * N, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
      }
      meta_host[(size_t)(j0 + j) * 4 + 0] = lml[j];
      meta_host[(size_t)(j0 + j) * 4 + 1] = jit[j];
      meta_host[(size_t)(j0 + j) * 4 + 2] = (double)info[j];
      meta_host[(size_t)(j0 + j) * 4 + 3] = 1.0;
    }
    if (simulate) HIPCHK(hipStreamSynchronize(c->stream));
  }
  return 0;
}

static int fit_sharded(bocf_ctx* c, const double* X, const double* Y, int N, int d, int m, int kernel_id, const double* variance,
                       const double* lengthscale, const double* noise, int max_jitter_tries, double* jitter_out, double* lml_out) {
  HIPCHK(hipSetDevice(c->device));
  const int simulate = c->shard_fit_simulate;                 // test hook (BOCF_PROBES builds): one process plays all G ranks in turn, no collectives
  const int G = simulate > 0 ? simulate : (c->comm ? c->world : 1), me = simulate > 0 ? 0 : (c->comm ? c->rank : 0);
  c->fitted = false; c->canned = false; c->have_acq = false; c->r32_valid = false;
  const int Np = round_up(N, BOCF_TILE), nb = Np / BOCF_TILE;
  c->N = N; c->Np = Np; c->d = d; c->m = m; c->kernel_id = kernel_id;
  const long strideS = (long)Np * Np;
  c->xs_stride = (long)Np * d;
  const size_t meta_w = (size_t)Np + N + 4;                   // alpha | train mean | lml, jitter, info, owner-count
  const size_t meta_n = (size_t)m * meta_w + 2;               // + number of ranks whose local share failed (+ padding)
  const size_t tiles = (size_t)nb * (nb + 1) / 2, packed = tiles * BOCF_TILE * BOCF_TILE;    // the exchanged part of one inverse factor
  // ---- local phase.  From here to the collectives NOTHING returns: a rank that left early would leave its peers blocked in
  // ncclAllReduce / ncclBroadcast for ever.  A local failure travels in the last slot of the meta block instead, and
  // every rank fails together after the exchange.
  int local_rc = 0;
  std::string local_err;
  std::vector<double> meta_host((size_t)m * 4, 0.0);
  auto local = [&]() -> int {
    if (c->X.ensure(sizeof(double) * (size_t)Np * d) || c->Xs.ensure(sizeof(double) * (size_t)m * Np * d) ||
        c->R.ensure(sizeof(double) * strideS * m) || c->RT.ensure(sizeof(double) * strideS * m) || c->yc.ensure(sizeof(double) * (size_t)m * Np) ||
        c->alpha.ensure(sizeof(double) * (size_t)m * Np) || c->lml.ensure(sizeof(double) * m) || c->hypd.ensure(sizeof(KernHyp) * m) ||
        c->mu_train.ensure(sizeof(double) * (size_t)m * Np) || c->meanpart.ensure(sizeof(double) * (size_t)2 * m * nb * Np) ||
        c->shard_meta.ensure(sizeof(double) * meta_n) || c->T.ensure(sizeof(double) * packed * m))
      return -1;
    HIPCHK(hipMemsetAsync(c->shard_meta.p, 0, sizeof(double) * meta_n, c->stream));
    // only the tiles on / above the diagonal of R are ever written (by the owner's fit or by the unpacking below): the other half
    // must be zeros, so a buffer that is new or was laid out for another size is cleared first
    if (c->zeroed_R != c->R.p || c->zeroed_Np != Np || c->zeroed_m < m) HIPCHK(hipMemsetAsync(c->R.p, 0, sizeof(double) * strideS * m, c->stream));
    if (stage_data(c, X, Y, N, Np, d, m, variance, lengthscale, noise)) return -1;
    launch_scale_inputs(c->X.as<double>(), N, d, c->hypd.as<KernHyp>(), m, c->Xs.as<double>(), c->xs_stride, c->stream);
    if (!c->shard_helper && bocf_create(c->device, &c->shard_helper)) return -1;
    bocf_ctx* hctx = c->shard_helper;
    // the helper factorizes with the caller's schedule: every schedule option is forwarded and the schedule is chosen for the
    // GLOBAL output count, so a share is factorized by the very kernel sequence the replicated fit would run for that output
    hctx->aggregate = c->aggregate; hctx->lookahead = c->lookahead; hctx->lookahead_min_nb = c->lookahead_min_nb;
    hctx->overlap_inverse = c->overlap_inverse; hctx->potrf_scalar = c->potrf_scalar; hctx->gemm_waves = c->gemm_waves;
    hctx->trsm_wave = c->trsm_wave; hctx->merge_x3 = c->merge_x3; hctx->gated_off = c->gated_off;
    hctx->sched_m = m;
    hctx->test_diag_shift = c->test_diag_shift;
    return fit_sharded_local(c, hctx, G, me, simulate, X, Y, N, d, m, kernel_id, variance, lengthscale, noise, max_jitter_tries, meta_w, meta_host);
  };
  local_rc = local();
  if (local_rc < 0) local_err = bocf_last_error();
  const bool have_meta = c->shard_meta.cap >= sizeof(double) * meta_n;
  const double failed_here = local_rc < 0 ? 1.0 : 0.0;
  if (have_meta) {
    for (int j = 0; j < m; ++j)
      (void)hipMemcpyAsync(c->shard_meta.as<double>() + (size_t)j * meta_w + Np + N, meta_host.data() + (size_t)j * 4, sizeof(double) * 4,
                           hipMemcpyHostToDevice, c->stream);
    (void)hipMemcpyAsync(c->shard_meta.as<double>() + (size_t)m * meta_w, &failed_here, sizeof(double), hipMemcpyHostToDevice, c->stream);
  }
  // ---- exchange (every rank, unconditionally): the small vectors by ONE all-reduce(SUM) (one owner per element), the inverse
  // factors by one broadcast per output from its owner, all in one group.  Only what prediction reads travels: the tiles on and
  // above the diagonal of R (nb (nb + 1) / 2 of the nb^2 tiles: 67 instead of 134 MB per output at N = 4096), packed
  // into T and unpacked after the exchange; the strictly lower part of R is the zero half no fit ever writes.
  int comm_rc = 0;
  double failed_ranks = failed_here;
  if (!simulate && c->comm && G > 1) {
    const bool can_take_part = have_meta && c->R.cap >= sizeof(double) * strideS * m && c->T.cap >= sizeof(double) * packed * m;
    if (!can_take_part) {
      // this rank could not even allocate the exchange buffers: it cannot issue collectives of the agreed sizes, so it ABORTS the
      // communicator -- the peers' collectives then fail with an RCCL error instead of waiting for ever
      (void)bocf_comm_abort(c);
      comm_rc = -1;
    }
    if (comm_rc == 0) {
      int j0m, j1m;
      shard_range(m, G, me, &j0m, &j1m);
      for (int j = j0m; j < j1m; ++j)
        launch_pack_upper_tiles(c->R.as<double>() + (size_t)j * strideS, Np, c->T.as<double>() + (size_t)j * packed, c->stream);
      if (bocf_comm_allreduce_sum(c, c->shard_meta.as<double>(), meta_n)) comm_rc = -1;
      if (bocf_comm_group(true)) comm_rc = -1;
      for (int r = 0; r < G && comm_rc == 0; ++r) {
        int j0, j1;
        shard_range(m, G, r, &j0, &j1);
        for (int j = j0; j < j1; ++j)
          if (bocf_comm_broadcast(c, c->T.as<double>() + (size_t)j * packed, packed, r)) comm_rc = -1;
      }
      if (bocf_comm_group(false)) comm_rc = -1;
      if (comm_rc == 0) {
        for (int j = 0; j < m; ++j)
          if (j < j0m || j >= j1m)
            launch_unpack_upper_tiles(c->T.as<double>() + (size_t)j * packed, Np, c->R.as<double>() + (size_t)j * strideS, c->stream);
        (void)hipMemcpyAsync(&failed_ranks, c->shard_meta.as<double>() + (size_t)m * meta_w, sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess) comm_rc = -1;
      }
    }
  }
  if (local_rc < 0) return fail("bocf_fit (sharded): this rank's share failed", local_err.c_str());
  if (comm_rc < 0) return -1;
  if (failed_ranks > 0.0) return fail("bocf_fit (sharded)", "another rank failed in its share of the outputs (its own error names the cause)");
  // unpack the small vectors, rebuild R^T
  std::vector<double> tail((size_t)m * 4);
  for (int j = 0; j < m; ++j) {
    const double* row = c->shard_meta.as<double>() + (size_t)j * meta_w;
    HIPCHK(hipMemcpyAsync(c->alpha.as<double>() + (size_t)j * Np, row, sizeof(double) * Np, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->mu_train.as<double>() + (size_t)j * N, row + Np, sizeof(double) * N, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(tail.data() + (size_t)j * 4, row + Np + N, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
  }
  launch_transpose_block(c->R.as<double>(), c->RT.as<double>(), strideS, Np, 0, 0, Np, Np, 1, 0, m, c->stream);
  HIPCHK(hipStreamSynchronize(c->stream));
  LAUNCHCHK();
  c->jitter.assign(m, 0.0);
  c->last_info.assign(m, 0);
  int bad = 0;
  std::vector<double> lml(m);
  for (int j = 0; j < m; ++j) {
    if (tail[(size_t)j * 4 + 3] != 1.0) return fail("bocf_fit (sharded)", "an output was factorized by no rank or by several");
    lml[j] = tail[(size_t)j * 4];
    c->jitter[j] = tail[(size_t)j * 4 + 1];
    c->last_info[j] = (int)tail[(size_t)j * 4 + 2];
    if (c->last_info[j] != 0 && bad == 0) bad = c->last_info[j];
  }
  if (jitter_out) memcpy(jitter_out, c->jitter.data(), sizeof(double) * m);
  if (bad) {
    bocf_fail("bocf_fit", "not positive definite, even with jitter.");
    return bad;
  }
  HIPCHK(hipMemcpyAsync(c->lml.p, lml.data(), sizeof(double) * m, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (lml_out) memcpy(lml_out, lml.data(), sizeof(double) * m);
  // R keeps zeros below its diagonal tiles, the transpose wrote all of R^T (zeros above): both are in the layout bocf_fit expects
  c->zeroed_R = c->R.p; c->zeroed_RT = c->RT.p; c->zeroed_Np = Np; c->zeroed_m = m;
  c->sharded = true;
  c->fitted = true;
  return 0;
}

extern "C" int bocf_fit(bocf_ctx* c, const double* X, const double* Y, int N, int d, int m, int kernel_id, const double* variance,
                        const double* lengthscale, const double* noise, int max_jitter_tries, double* jitter_out, double* lml_out) {
  if (!c || !X || !Y || !variance || !lengthscale || !noise) return fail("bocf_fit", "null argument");
  if (N < 1 || d < 1 || d > BOCF_MAX_D || m < 1 || m > BOCF_MAX_FITS) return fail("bocf_fit", "N, d or m out of range");
  if (kernel_id < 0 || kernel_id > 3) return fail("bocf_fit", "unknown kernel id");
  for (int j = 0; j < m; ++j) {
    if (!(variance[j] > 0.0) || !(noise[j] >= 0.0)) return fail("bocf_fit", "variance must be > 0 and noise >= 0");
    for (int q = 0; q < d; ++q)
      if (!(lengthscale[(long)j * d + q] > 0.0)) return fail("bocf_fit", "lengthscale must be > 0");
  }
  if ((c->shard_fit && (c->comm || c->shard_fit_simulate > 0)) && !c->reuse_data && m > 1 && m % c->hyper_samples == 0)
    return fit_sharded(c, X, Y, N, d, m, kernel_id, variance, lengthscale, noise, max_jitter_tries, jitter_out, lml_out);
  HIPCHK(hipSetDevice(c->device));
  c->fitted = false;
  c->canned = false;
  c->sharded = false;
  c->have_acq = false;
  c->r32_valid = false;
  const int Np = round_up(N, BOCF_TILE), nb = Np / BOCF_TILE;
  c->N = N; c->Np = Np; c->d = d; c->m = m; c->kernel_id = kernel_id;
  const long strideS = (long)Np * Np;
  c->xs_stride = (long)Np * d;
  if (c->X.ensure(sizeof(double) * (size_t)Np * d) || c->Xs.ensure(sizeof(double) * (size_t)m * Np * d) ||
      c->S.ensure(sizeof(double) * strideS * m) || c->R.ensure(sizeof(double) * strideS * m) ||
      c->E.ensure(sizeof(double) * (size_t)m * nb * BOCF_TILE * BOCF_TILE) ||
      c->ET.ensure(sizeof(double) * (size_t)m * nb * BOCF_TILE * BOCF_TILE) ||
      c->T.ensure(sizeof(double) * strideS * m) || c->RT.ensure(sizeof(double) * strideS * m) || c->yc.ensure(sizeof(double) * (size_t)m * Np) ||
      c->tvec.ensure(sizeof(double) * (size_t)m * Np) || c->alpha.ensure(sizeof(double) * (size_t)m * Np) ||
      c->lml.ensure(sizeof(double) * m) || c->jit.ensure(sizeof(double) * m) || c->hypd.ensure(sizeof(KernHyp) * m) ||
      c->info.ensure(sizeof(int) * m) || c->mu_train.ensure(sizeof(double) * (size_t)m * Np) ||
      c->meanpart.ensure(sizeof(double) * (size_t)2 * m * nb * Np))
    return -1;

  if (stage_data(c, X, Y, N, Np, d, m, variance, lengthscale, noise)) return -1;
  launch_scale_inputs(c->X.as<double>(), N, d, c->hypd.as<KernHyp>(), m, c->Xs.as<double>(), c->xs_stride, c->stream);
  // R (upper) and R^T (lower) are rewritten block by block by every fit; their other triangles are zeros that nothing ever
  // writes, so they are cleared only when the buffers are new or laid out for another padded size
  if (c->zeroed_R != c->R.p || c->zeroed_RT != c->RT.p || c->zeroed_Np != Np || c->zeroed_m < m) {
    HIPCHK(hipMemsetAsync(c->R.p, 0, sizeof(double) * strideS * m, c->stream));
    HIPCHK(hipMemsetAsync(c->RT.p, 0, sizeof(double) * strideS * m, c->stream));
    c->zeroed_R = c->R.p; c->zeroed_RT = c->RT.p; c->zeroed_Np = Np; c->zeroed_m = m;
  }
  if (c->overlap_inverse != 0 && !c->s_inv) {
    // the early part of the inverse runs on its own stream; where the runtime allows CU masks it keeps off the CUs the
    // diagonal-block kernel of the (unmasked) main stream then finds free
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    const int keep = 32;       // 4 CUs of every XCD: the row products of the chain (hundreds of small workgroups) need more room than the diagonal
                               // blocks alone; measured 8 / 32 / 64 / 96 kept: 7.06 / 7.00 / 6.98 / 7.00 ms at config 3, 32.9 / 32.0 / 32.4 at N = 8192
    std::vector<uint32_t> mask(words, 0u);
    for (int i = keep < ncu / 2 ? keep : 0; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
    if (c->cu_masks_ok && hipExtStreamCreateWithCUMask(&c->s_inv, (uint32_t)words, mask.data()) != hipSuccess) {
      (void)hipGetLastError();
      c->s_inv = nullptr;
      c->cu_masks_ok = 0;
    }
    if (!c->s_inv) HIPCHK(hipStreamCreate(&c->s_inv));
    HIPCHK(hipEventCreateWithFlags(&c->ev_half, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_inv_early, hipEventDisableTiming));
  }

  // jitchol ladder (GPy/util/linalg.py:52-71)
  c->jitter.assign(m, 0.0);
  std::vector<int> info(m, 0);
  int bad = 0;
  for (int attempt = 0;; ++attempt) {
    std::vector<double> jeff(c->jitter);
    for (int j = 0; j < m; ++j) jeff[j] -= c->test_diag_shift;
    HIPCHK(hipMemcpyAsync(c->jit.p, jeff.data(), sizeof(double) * m, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->info.p, 0, sizeof(int) * m, c->stream));
    {
      PhaseTimer t(c, "kbuild");
      launch_build_train_kernel(c->Xs.as<double>(), c->xs_stride, N, Np, d, kernel_id, c->hypd.as<KernHyp>(), c->jit.as<double>(), 1,
                                c->S.as<double>(), strideS, m, c->stream);
    }
    {
      PhaseTimer t(c, "cholesky");
      if (run_cholesky(c)) return -1;
      // (a failed attempt is rebuilt from scratch: the early inverse must be off the buffers first -- the wait costs nothing
      //  when the attempt succeeded, the inverse phase would wait for the same event)
      if (c->early_inverse_started) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_inv_early, 0));
    }
    HIPCHK(hipMemcpyAsync(info.data(), c->info.p, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
    int sched_err = 0;
    if (c->chol_flags_used)
      HIPCHK(hipMemcpyAsync(&sched_err, c->chol_flags.as<int>() + 5 * nb, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
#ifdef BOCF_PROBES
    if (c->force_sched_timeout && c->chol_flags_used) {      // test hook: as if a gate had run out of polls
      sched_err = 1;
      c->force_sched_timeout = 0;
    }
#endif
    c->chol_flags_used = 0;
    if (sched_err && getenv("BOCF_DBG_FLAGS")) {             // which counters had arrived when the time-out fired
      std::vector<int> fl(c->chol_flags.cap / sizeof(int));
      (void)hipMemcpy(fl.data(), c->chol_flags.p, fl.size() * sizeof(int), hipMemcpyDeviceToHost);
      fprintf(stderr, "bocf_fit: dependency time-out (schedule %d, nb %d), first wait that ran out: id %d\n", c->last_schedule, nb, sched_err);
      chol_chain_dbg_dump();
    }
    if (sched_err) {
      // A gate of a multi-stream schedule ran out of polls (0.2 s): its consumers ran on incomplete tiles.  That depends on timing
      // (a host stall while the streams are being filled, a tool that serialises dispatches across queues), not on the data:
      // rebuild K and redo THIS attempt on the single-stream schedule (c->sched_retry), count it; from the second time on the
      // gated schedules stay off for the context.
      c->sched_timeouts++;
      if (c->sched_timeouts >= 2) c->gated_off = 1;          // once may be a one-time stall (first use of a code object, a descheduled host thread); twice is a pattern
      if (c->sched_timeouts > 8) return fail("bocf_fit", "the factorization schedule keeps timing out waiting for device-side dependencies");
      c->sched_retry = 1;
      --attempt;
      continue;
    }
    bad = 0;
    for (int j = 0; j < m; ++j)
      if (info[j] != 0 && bad == 0) bad = info[j];
    if (!bad) break;
    if (attempt >= max_jitter_tries) break;
    for (int j = 0; j < m; ++j)
      if (info[j] != 0) {
        const double diag_mean = c->hyp[j].variance + c->hyp[j].noise + 1e-8 - c->test_diag_shift;   // mean(diag(Ky)), stationary kernel
        c->jitter[j] = c->jitter[j] == 0.0 ? diag_mean * 1e-6 : c->jitter[j] * 10.0;
      }
  }
  if (jitter_out) memcpy(jitter_out, c->jitter.data(), sizeof(double) * m);
  c->last_info = info;
  if (bad) {
    g_err = "not positive definite, even with jitter.";
    return bad;
  }
  {
    PhaseTimer t(c, "inverse");
    if (c->early_inverse_started) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_inv_early, 0));
    if (run_trtri(c, c->early_inverse_started != 0)) return -1;
  }
  PhaseTimer t_alpha(c, "alpha");
  if (solve_alpha(c, !c->skip_mu_train)) return -1;
  t_alpha.stop();
  if (lml_out) HIPCHK(hipMemcpyAsync(lml_out, c->lml.p, sizeof(double) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  LAUNCHCHK();
  c->fitted = true;
  c->fits_done++;
  return 0;
}

// yc, alpha, log-marginal and the cached posterior mean at the training inputs from host targets Y (m, N)
static int refresh_targets(bocf_ctx* c, const double* Y, double* lml_out) {
  const int N = c->N, Np = c->Np, m = c->m, d = c->d, nb = Np / BOCF_TILE;
  const long strideS = (long)Np * Np;
  std::vector<double> yc((size_t)m * Np, 0.0);
  for (int j = 0; j < m; ++j) {
    double s = 0.0;
    for (int i = 0; i < N; ++i) s += Y[(long)j * N + i];
    const double mean = s / N;
    c->hyp[j].ymean = mean;
    for (int i = 0; i < N; ++i) yc[(long)j * Np + i] = Y[(long)j * N + i] - mean;
  }
  HIPCHK(hipMemcpyAsync(c->hypd.p, c->hyp.data(), sizeof(KernHyp) * m, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->yc.p, yc.data(), sizeof(double) * (size_t)m * Np, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (solve_alpha(c, true)) return -1;
  if (lml_out) HIPCHK(hipMemcpyAsync(lml_out, c->lml.p, sizeof(double) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  LAUNCHCHK();
  return 0;
}

extern "C" int bocf_update_targets(bocf_ctx* c, const double* Y, double* lml_out) {
  if (!c || !c->fitted || c->canned || !Y) return fail("bocf_update_targets", "model not fitted / null Y");
  if (c->sharded) return fail("bocf_update_targets", "the fit is output-sharded: refit");
  HIPCHK(hipSetDevice(c->device));
  c->have_acq = false;
  return refresh_targets(c, Y, lml_out);
}

extern "C" int bocf_append(bocf_ctx* c, const double* x_new, const double* Y, double* lml_out) {
  if (!c || !c->fitted || c->canned || !x_new || !Y) return fail("bocf_append", "model not fitted / null argument");
  if (c->sharded) return 1;                              // an output-sharded fit keeps no upper factor to border: the caller refits
  HIPCHK(hipSetDevice(c->device));
  const int N = c->N, Np = c->Np, m = c->m, d = c->d, nb = Np / BOCF_TILE;
  if (N >= Np) return 1;                               // no padding row left: the caller refits
  for (int j = 0; j < m; ++j)
    if (c->jitter[j] != 0.0) return 1;                 // a jittered factor is not extended (the ladder decides from scratch)
  const long strideS = (long)Np * Np, strideE = (long)nb * BOCF_TILE * BOCF_TILE;
  c->have_acq = false;
  if (c->Xc.ensure(sizeof(double) * d) || c->Kstar.ensure(sizeof(double) * (size_t)m * Np * BOCF_TILE) ||
      c->sumsq.ensure(sizeof(double) * (size_t)m * BOCF_TILE) || c->Vs.ensure(sizeof(double) * (size_t)m * Np * BOCF_SMALL_N) ||
      c->Ws.ensure(sizeof(double) * (size_t)m * Np * BOCF_SMALL_N) || c->meanpart.ensure(sizeof(double) * (size_t)2 * m * nb * Np))
    return -1;
  c->C = 0;                                            // the resident candidate batch is replaced
  HIPCHK(hipMemcpyAsync(c->Xc.p, x_new, sizeof(double) * d, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(c->info.p, 0, sizeof(int) * m, c->stream));
  // k(X, x_new) as column 0 of a 128-wide K* block, u = R^T k, ||u||^2, w = R u
  launch_cross_kernel(c->Xs.as<double>(), (long)c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->Xc.as<double>(), 0, 1, BOCF_TILE,
                      c->alpha.as<double>(), c->Kstar.as<double>(), BOCF_TILE, (long)Np * BOCF_TILE, c->meanpart.as<double>(),
                      c->meanpart.as<double>() + (size_t)m * nb * Np, 1, m, 1, c->stream);
  launch_gemv_small_t(c->R.as<double>(), strideS, Np, c->Kstar.as<double>(), BOCF_TILE, (long)Np * BOCF_TILE, c->Vs.as<double>(), 1, m, c->stream);
  launch_sumsq_small(c->Vs.as<double>(), Np, c->sumsq.as<double>(), BOCF_TILE, 1, m, c->stream);
  launch_gemv_small_n(c->R.as<double>(), strideS, Np, c->Vs.as<double>(), c->Ws.as<double>(), 1, m, c->stream);
  launch_append_write(c->S.as<double>(), c->R.as<double>(), c->RT.as<double>(), strideS, c->E.as<double>(), c->ET.as<double>(), strideE, Np, N,
                      c->Vs.as<double>(), c->Ws.as<double>(), c->sumsq.as<double>(), BOCF_TILE, c->hypd.as<KernHyp>(), c->info.as<int>(), m,
                      c->stream);
  std::vector<int> failed(m, 0);
  HIPCHK(hipMemcpyAsync(failed.data(), c->info.p, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int j = 0; j < m; ++j)
    if (failed[j]) {
      c->fitted = false;                               // some outputs may already be extended: force a full refit
      return 1;
    }
  // the new input joins X / Xs (row N of the per-output blocks, which are laid out with capacity Np)
  HIPCHK(hipMemcpyAsync(c->X.as<double>() + (size_t)N * d, x_new, sizeof(double) * d, hipMemcpyHostToDevice, c->stream));
  launch_scale_inputs(c->X.as<double>() + (size_t)N * d, 1, d, c->hypd.as<KernHyp>(), m, c->Xs.as<double>() + (size_t)N * d, c->xs_stride,
                      c->stream);
  c->N = N + 1;
  c->r32_valid = false;
  if (c->mu_train.ensure(sizeof(double) * (size_t)m * c->N)) return -1;
  return refresh_targets(c, Y, lml_out);
}

extern "C" int bocf_lml_gradients(bocf_ctx* c, double* dvariance_out, double* dlengthscale_out, double* dnoise_out) {
  if (!c || !c->fitted || c->canned) return fail("bocf_lml_gradients", "model not fitted");
  if (c->sharded) return fail("bocf_lml_gradients", "the fit is output-sharded (hyper-parameter learning factorizes unsharded)");
  HIPCHK(hipSetDevice(c->device));
  const int N = c->N, Np = c->Np, m = c->m, d = c->d;
  const long strideS = (long)Np * Np;
  const int nblk = hypgrad_num_blocks(Np);
  DevBuf& part = c->gpart;
  DevBuf& out = c->gout;
  if (part.ensure(sizeof(double) * (size_t)m * nblk * (2 + d)) || out.ensure(sizeof(double) * (size_t)m * (2 + d))) return -1;
  // Ky^-1 = R R^T, upper tiles: Kinv[r][c] = sum_{kk >= max(r,c)} RT[kk][r] RT[kk][c]   (into the T scratch)
  GemmArgs g{};
  g.A = c->RT.as<double>(); g.lda = Np; g.strideA = strideS;
  g.B = c->RT.as<double>(); g.ldb = Np; g.strideB = strideS;
  g.Cin = nullptr; g.Cout = c->T.as<double>(); g.ldc = Np; g.strideC = strideS;
  g.M = Np; g.Ncols = Np; g.K = Np; g.kb = Np; g.kbeg_ct = BOCF_TILE; g.upper_only = 1; g.alpha = 1.0;
  launch_gemm_f64(g, m, 0, c->stream);
  launch_hypgrad(c->Xs.as<double>(), c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->alpha.as<double>(), c->T.as<double>(),
                 strideS, part.as<double>(), out.as<double>(), m, c->stream);
  std::vector<double> h((size_t)m * (2 + d));
  hipError_t e = hipMemcpyAsync(h.data(), out.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail("bocf_lml_gradients", hipGetErrorString(e));
  for (int j = 0; j < m; ++j) {
    if (dvariance_out) dvariance_out[j] = h[(size_t)j * (2 + d)];
    if (dnoise_out) dnoise_out[j] = h[(size_t)j * (2 + d) + 1];
    if (dlengthscale_out)
      for (int q = 0; q < d; ++q) dlengthscale_out[(size_t)j * d + q] = h[(size_t)j * (2 + d) + 2 + q];
  }
  LAUNCHCHK();
  return 0;
}

// One hyper-parameter inference: log-marginal and its gradients at the given hyper-parameters -- the unit of work of
// GPModel.updateModel's optimiser and HMC (gpmodel.py:115-118; hmc.py:62-66 calls it 20 times per draw).  Models with
// N <= 128 and d <= 16 (the usual size of a BO run) take ONE fused launch per jitter attempt; anything else is
// bocf_fit + bocf_lml_gradients.  The fused path leaves no factor behind (the context is un-fitted afterwards).
extern "C" int bocf_infer(bocf_ctx* c, const double* X, const double* Y, int N, int d, int m, int kernel_id, const double* variance,
                          const double* lengthscale, const double* noise, int max_jitter_tries, double* jitter_out, double* lml_out,
                          double* dvariance_out, double* dlengthscale_out, double* dnoise_out) {
  if (!c || !X || !Y || !variance || !lengthscale || !noise) return fail("bocf_infer", "null argument");
  const int Np = round_up(N < 1 ? 1 : N, BOCF_TILE);
  if (!c->fused_infer || Np != BOCF_TILE || d > BOCF_INFER_MAX_D) {
    const int sf = c->shard_fit;                         // an inference needs the upper factor on this rank: never output-sharded
    c->shard_fit = 0;
    const int rc = bocf_fit(c, X, Y, N, d, m, kernel_id, variance, lengthscale, noise, max_jitter_tries, jitter_out, lml_out);
    c->shard_fit = sf;
    if (rc) return rc;
    return bocf_lml_gradients(c, dvariance_out, dlengthscale_out, dnoise_out);
  }
  if (N < 1 || d < 1 || m < 1 || m > BOCF_MAX_FITS) return fail("bocf_infer", "N, d or m out of range");
  if (kernel_id < 0 || kernel_id > 3) return fail("bocf_infer", "unknown kernel id");
  for (int j = 0; j < m; ++j) {
    if (!(variance[j] > 0.0) || !(noise[j] >= 0.0)) return fail("bocf_infer", "variance must be > 0 and noise >= 0");
    for (int q = 0; q < d; ++q)
      if (!(lengthscale[(long)j * d + q] > 0.0)) return fail("bocf_infer", "lengthscale must be > 0");
  }
  HIPCHK(hipSetDevice(c->device));
  c->fitted = false;
  c->canned = false;
  c->have_acq = false;
  c->r32_valid = false;
  c->N = N; c->Np = Np; c->d = d; c->m = m; c->kernel_id = kernel_id;
  const int nout = 2 + d + 2;                                // gradients, log-marginal, info
  if (c->X.ensure(sizeof(double) * (size_t)Np * d) || c->yc.ensure(sizeof(double) * (size_t)m * Np) || c->hypd.ensure(sizeof(KernHyp) * m))
    return -1;
  if (c->infer_out_cap < (size_t)m * nout) {
    if (c->infer_out) (void)hipHostFree(c->infer_out);
    c->infer_out = nullptr;
    c->infer_out_cap = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->infer_out), sizeof(double) * (size_t)m * nout, hipHostMallocMapped));
    c->infer_out_cap = (size_t)m * nout;
  }
  double* out_dev = nullptr;
  HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&out_dev), c->infer_out, 0));
  if (stage_data(c, X, Y, N, Np, d, m, variance, lengthscale, noise)) return -1;
  c->jitter.assign(m, 0.0);
  std::vector<int> info(m, 0);
  std::vector<double> lml(m), out((size_t)m * nout);
  int bad = 0;
  for (int attempt = 0;; ++attempt) {                      // jitchol ladder (GPy/util/linalg.py:52-71)
    if (attempt > 0) {                                     // attempt 0: stage_data uploaded the hyper-parameters with jitter 0
      for (int j = 0; j < m; ++j) c->hyp[j].jitter = c->jitter[j] - c->test_diag_shift;
      HIPCHK(hipMemcpyAsync(c->hypd.p, c->hyp.data(), sizeof(KernHyp) * m, hipMemcpyHostToDevice, c->stream));
    }
    launch_infer128(c->X.as<double>(), N, d, kernel_id, c->hypd.as<KernHyp>(), c->yc.as<double>(), out_dev, m, c->stream);
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(out.data(), c->infer_out, sizeof(double) * out.size());
    bad = 0;
    for (int j = 0; j < m; ++j) {
      info[j] = (int)out[(size_t)j * nout + nout - 1];
      lml[j] = out[(size_t)j * nout + nout - 2];
      if (info[j] != 0 && bad == 0) bad = info[j];
    }
    if (!bad || attempt >= max_jitter_tries) break;
    for (int j = 0; j < m; ++j)
      if (info[j] != 0) {
        const double diag_mean = c->hyp[j].variance + c->hyp[j].noise + 1e-8 - c->test_diag_shift;
        c->jitter[j] = c->jitter[j] == 0.0 ? diag_mean * 1e-6 : c->jitter[j] * 10.0;
      }
  }
  LAUNCHCHK();
  if (jitter_out) memcpy(jitter_out, c->jitter.data(), sizeof(double) * m);
  c->last_info = info;
  if (bad) {
    g_err = "not positive definite, even with jitter.";
    return bad;
  }
  for (int j = 0; j < m; ++j) {
    if (lml_out) lml_out[j] = lml[j];
    if (dvariance_out) dvariance_out[j] = out[(size_t)j * nout];
    if (dnoise_out) dnoise_out[j] = out[(size_t)j * nout + 1];
    if (dlengthscale_out)
      for (int q = 0; q < d; ++q) dlengthscale_out[(size_t)j * d + q] = out[(size_t)j * nout + 2 + q];
  }
  return 0;
}

// The whole HMC chain of GPModel.updateModel (gpmodel.py:117-118 -> GPy/inference/mcmc/hmc.py:30-69) on the device, for models the
// fused inference serves (N <= 128, d <= 16): ONE launch, one workgroup per output, every leapfrog step an in-kernel inference (with
// jitchol's ladder) plus the O(P) transform / prior / momentum arithmetic; the host only draws the momenta and uniforms (in the
// reference's RNG order) and reads the chains back.  Returns 0, or 1 when some output's chain stopped on a failed factorization with
// raise_on_failure (status_out says which, and in which draw), or < 0.
extern "C" int bocf_hmc(bocf_ctx* c, const double* X, const double* Y, int N, int d, int m, int kernel_id, double* theta, int nls,
                        const int* fixed, double prior_a, double prior_b, const double* momenta, const double* uniforms, int num_samples,
                        int hmc_iters, double stepsize, int max_jitter_tries, int raise_on_failure, double* chains_out, int* accepted_out,
                        int* diverged_out, int* status_out, long long* inferences_out) {
  if (!c || !X || !Y || !theta || !fixed || !momenta || !uniforms || !chains_out || !accepted_out || !status_out)
    return fail("bocf_hmc", "null argument");
  if (N < 1 || N > BOCF_TILE || d < 1 || d > BOCF_INFER_MAX_D || m < 1 || m > BOCF_MAX_FITS) return fail("bocf_hmc", "N (<= 128), d (<= 16) or m out of range");
  if (kernel_id < 0 || kernel_id > 3) return fail("bocf_hmc", "unknown kernel id");
  if (nls != 1 && nls != d) return fail("bocf_hmc", "nls must be 1 (isotropic) or d (ARD)");
  if (num_samples < 1 || hmc_iters < 1 || !(stepsize > 0.0) || !(prior_a > 0.0) || !(prior_b > 0.0) || max_jitter_tries < 0)
    return fail("bocf_hmc", "num_samples, hmc_iters, stepsize, prior or max_jitter_tries out of range");
  const int P = 2 + nls, Np = BOCF_TILE;
  for (int j = 0; j < m; ++j) {
    int nfree = 0;
    for (int k = 0; k < P; ++k) {
      const double t = theta[(size_t)j * P + k];
      if (!(t > 0.0) && !(k == P - 1 && t == 0.0)) return fail("bocf_hmc", "theta must be positive (noise >= 0)");
      nfree += fixed[(size_t)j * P + k] ? 0 : 1;
    }
    if (nfree < 1) return fail("bocf_hmc", "an output has no free parameter");
  }
  HIPCHK(hipSetDevice(c->device));
  c->fitted = false; c->canned = false; c->have_acq = false; c->r32_valid = false;
  c->N = N; c->Np = Np; c->d = d; c->m = m; c->kernel_id = kernel_id;
  if (c->X.ensure(sizeof(double) * (size_t)Np * d) || c->yc.ensure(sizeof(double) * (size_t)m * Np) || c->hypd.ensure(sizeof(KernHyp) * m)) return -1;
  {
    std::vector<double> var(m), ls((size_t)m * d), nz(m);
    for (int j = 0; j < m; ++j) {
      var[j] = theta[(size_t)j * P];
      nz[j] = theta[(size_t)j * P + P - 1];
      for (int q = 0; q < d; ++q) ls[(size_t)j * d + q] = theta[(size_t)j * P + 1 + (nls == 1 ? 0 : q)];
    }
    const int reuse = c->reuse_data;
    if (reuse && !(c->data_N == N && c->data_d == d && c->data_m == m && (int)c->hyp.size() == m)) c->reuse_data = 0;   // (first call of a data set)
    const int rc = stage_data(c, X, Y, N, Np, d, m, var.data(), ls.data(), nz.data());
    c->reuse_data = reuse;
    if (rc) return -1;
  }
  const size_t nth = (size_t)m * P, nmom = (size_t)m * num_samples * P, nuni = (size_t)m * num_samples;
  // one scratch block: theta | momenta | uniforms | chains (doubles), then fixed | accepted | diverged | status (ints), n_infer (long long)
  const size_t dbl = nth + nmom + nuni + nmom, ints = nth + 3 * (size_t)m;
  const size_t bytes = sizeof(double) * dbl + sizeof(long long) * m + sizeof(int) * ints;
  if (c->hmc_buf.ensure(bytes)) return -1;
  double* dth = c->hmc_buf.as<double>();
  double* dmom = dth + nth;
  double* duni = dmom + nmom;
  double* dch = duni + nuni;
  long long* dninf = reinterpret_cast<long long*>(dch + nmom);
  int* dfix = reinterpret_cast<int*>(dninf + m);
  int* dacc = dfix + nth;
  int* ddiv = dacc + m;
  int* dst = ddiv + m;
  HIPCHK(hipMemcpyAsync(dth, theta, sizeof(double) * nth, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dmom, momenta, sizeof(double) * nmom, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(duni, uniforms, sizeof(double) * nuni, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dfix, fixed, sizeof(int) * nth, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(dch, 0, sizeof(double) * nmom, c->stream));
  HIPCHK(hipMemsetAsync(dacc, 0, sizeof(int) * 3 * (size_t)m, c->stream));
  HmcArgs a{};
  a.X = c->X.as<double>(); a.N = N; a.d = d; a.yc = c->yc.as<double>();
  a.theta = dth; a.fixed = dfix; a.P = P; a.nls = nls;
  a.prior_a = prior_a; a.prior_b = prior_b; a.prior_const = -lgamma(prior_a) + prior_a * log(prior_b);   // priors.py:271
  a.mom = dmom; a.uni = duni; a.ns = num_samples; a.iters = hmc_iters; a.eps = stepsize;
  a.max_tries = max_jitter_tries; a.raise_on_failure = raise_on_failure ? 1 : 0; a.diag_shift = c->test_diag_shift;
  a.chains = dch; a.accepted = dacc; a.diverged = ddiv; a.status = dst; a.n_infer = dninf;
  launch_hmc128(a, kernel_id, m, c->stream);
  std::vector<long long> ninf(m, 0);
  std::vector<int> dv(m, 0);
  HIPCHK(hipMemcpyAsync(theta, dth, sizeof(double) * nth, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(chains_out, dch, sizeof(double) * nmom, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(accepted_out, dacc, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(dv.data(), ddiv, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(status_out, dst, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(ninf.data(), dninf, sizeof(long long) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  LAUNCHCHK();
  if (diverged_out) memcpy(diverged_out, dv.data(), sizeof(int) * m);
  long long total = 0;
  int bad = 0;
  for (int j = 0; j < m; ++j) {
    total = ninf[j] > total ? ninf[j] : total;            // (the chains run side by side: one "batched inference" per step, as hyper.py counts)
    if (status_out[j] != 0) bad = 1;
  }
  if (inferences_out) *inferences_out = total;
  if (bad) g_err = "not positive definite, even with jitter.";
  return bad;
}

extern "C" int bocf_last_fit_info(bocf_ctx* c, int* info_out, int n) {
  if (!c || !info_out || n < 0) return fail("bocf_last_fit_info", "null argument");
  if ((size_t)n != c->last_info.size()) return fail("bocf_last_fit_info", "n differs from the number of outputs of the last fit");
  memcpy(info_out, c->last_info.data(), sizeof(int) * (size_t)n);
  return 0;
}

extern "C" int bocf_get_factor(bocf_ctx* c, int j, double* L_out, double* alpha_out) {
  if (!c || !c->fitted || c->canned) return fail("bocf_get_factor", "model not fitted");
  if (c->sharded && L_out) return fail("bocf_get_factor", "the fit is output-sharded: only the inverse factor is exchanged, L is not on this rank");
  if (j < 0 || j >= c->m) return fail("bocf_get_factor", "output index out of range");
  HIPCHK(hipSetDevice(c->device));
  const int N = c->N, Np = c->Np;
  if (L_out) {
    std::vector<double> S((size_t)Np * Np);
    HIPCHK(hipMemcpy(S.data(), c->S.as<double>() + (long)j * Np * Np, sizeof(double) * (size_t)Np * Np, hipMemcpyDeviceToHost));
    for (int r = 0; r < N; ++r)
      for (int cc = 0; cc < N; ++cc) L_out[(long)r * N + cc] = cc <= r ? S[(long)cc * Np + r] : 0.0;   // L = U^T, read from the upper factor (no mirrored copy is kept)
  }
  if (alpha_out) HIPCHK(hipMemcpy(alpha_out, c->alpha.as<double>() + (long)j * Np, sizeof(double) * N, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int bocf_get_train_kernel(bocf_ctx* c, int j, double* K_out) {
  if (!c || !c->fitted || c->canned || !K_out) return fail("bocf_get_train_kernel", "model not fitted / null out");
  if (j < 0 || j >= c->m) return fail("bocf_get_train_kernel", "output index out of range");
  HIPCHK(hipSetDevice(c->device));
  const int N = c->N, Np = c->Np;
  DevBuf tmp;
  if (tmp.ensure(sizeof(double) * (size_t)Np * Np)) return -1;
  launch_build_train_kernel(c->Xs.as<double>() + (long)j * c->xs_stride, 0, N, Np, c->d, c->kernel_id, c->hypd.as<KernHyp>() + j, nullptr, 0,
                            tmp.as<double>(), 0, 1, c->stream);
  std::vector<double> S((size_t)Np * Np);
  hipError_t e = hipMemcpyAsync(S.data(), tmp.p, sizeof(double) * (size_t)Np * Np, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  tmp.release();
  if (e != hipSuccess) return fail("bocf_get_train_kernel", hipGetErrorString(e));
  for (int r = 0; r < N; ++r)
    for (int cc = 0; cc < N; ++cc) K_out[(long)r * N + cc] = cc >= r ? S[(long)r * Np + cc] : S[(long)cc * Np + r];
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
  HIPCHK(hipStreamSynchronize(c->stream));
  c->m = m; c->N = N; c->Np = round_up(N, BOCF_TILE); c->d = 1; c->C = C; c->pred_cap = cap;
  c->fitted = true;
  c->canned = true;
  c->have_acq = false;
  c->S_mc = 0;
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
  HIPCHK(hipMemcpyAsync(c->Xc.p, Xc, sizeof(double) * (size_t)C * c->d, hipMemcpyHostToDevice, c->stream));
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
    if (c->Kstar.ensure(sizeof(double) * (size_t)m * Np * chunkpad) || c->sumsq.ensure(sizeof(double) * (size_t)m * nrt * chunkpad)) return -1;
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
      launch_cross_kernel(c->Xs.as<double>(), c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->Xc.as<double>(),
                          (int)c0 + pc0, pvalid, pcols, c->alpha.as<double>(), kbase, Cpad, (long)Np * Cpad,
                          c->meanpart.as<double>() + (size_t)pc0 * m * nrt, c->meanpart.as<double>() + mean_plane + (size_t)pc0 * m * nrt, ns, m,
                          need_var ? (f32 ? 2 : 1) : 0, sx);
      launch_finalize_mean(c->meanpart.as<double>() + (size_t)pc0 * m * nrt, c->meanpart.as<double>() + mean_plane + (size_t)pc0 * m * nrt, nrt,
                           pcols, c->hypd.as<KernHyp>(), c->mean.as<double>(), ld, (int)c0 + pc0, pvalid, m, sx);
      t_cross.stop();
      if (!need_var) continue;
      if (nparts > 1) {
        HIPCHK(hipEventRecord(c->ev_parts[part], c->stream2));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_parts[part], 0));
      }
      if (small) {
        // n <= 16: GEMV-shaped, R streamed once per product (single-point L-BFGS calls)
        int nc = 1;
        while (nc < Cn) nc *= 2;
        launch_gemv_small_t(c->R.as<double>(), strideS, Np, c->Kstar.as<double>(), Cpad, (long)Np * Cpad, c->Vs.as<double>(), nc, m, c->stream);
        launch_sumsq_small(c->Vs.as<double>(), Np, c->sumsq.as<double>(), Cpad, nc, m, c->stream);
        launch_finalize_var(c->sumsq.as<double>(), 1, Cpad, c->hypd.as<KernHyp>(), flags, c->var.as<double>(), ld, (int)c0, Cn, m, c->stream);
        if (need_grad) {
          launch_gemv_small_n(c->R.as<double>(), strideS, Np, c->Vs.as<double>(), c->Ws.as<double>(), nc, m, c->stream);
          launch_grad_kernel(c->Xs.as<double>(), c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->Xc.as<double>(), (int)c0, Cn,
                             c->alpha.as<double>(), c->Ws.as<double>(), nc, (long)Np * nc, c->dmean.as<double>(),
                             c->dvar.as<double>(), ld, m, c->stream);
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
                            (int)c0 + pc0, pvalid, m, c->stream);
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
                          (int)c0 + pc0, pvalid, m, c->stream);
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
                       m, c->stream);
  }
  LAUNCHCHK();
  return 0;
}

static int copy_rows_out(bocf_ctx* c, const double* dev, long ld, int rows, int C, double* out) {
  for (int j = 0; j < rows; ++j)
    HIPCHK(hipMemcpyAsync(out + (long)j * C, dev + (long)j * ld, sizeof(double) * C, hipMemcpyDeviceToHost, c->stream));
  return 0;
}

extern "C" int bocf_predict(bocf_ctx* c, int flags, double* mean_out, double* var_out) {
  if (!c || !c->fitted || c->canned) return fail("bocf_predict", "model not fitted");
  HIPCHK(hipSetDevice(c->device));
  if (c->C == 0) return 0;
  if (run_predict(c, flags, var_out != nullptr)) return -1;
  if (mean_out && copy_rows_out(c, c->mean.as<double>(), c->pred_cap, c->m, c->C, mean_out)) return -1;
  if (var_out && copy_rows_out(c, c->var.as<double>(), c->pred_cap, c->m, c->C, var_out)) return -1;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int bocf_predict_gradients(bocf_ctx* c, double* dmean_out, double* dvar_out) {
  if (!c || !c->fitted || c->canned) return fail("bocf_predict_gradients", "model not fitted");
  HIPCHK(hipSetDevice(c->device));
  if (c->C == 0) return 0;
  if (run_predict(c, BOCF_ADD_NOISE | BOCF_CLIP, true, true)) return -1;
  const size_t row = sizeof(double) * (size_t)c->C * c->d;
  for (int j = 0; j < c->m; ++j) {
    if (dmean_out)
      HIPCHK(hipMemcpyAsync(dmean_out + (size_t)j * c->C * c->d, c->dmean.as<double>() + (size_t)j * c->pred_cap * c->d, row,
                            hipMemcpyDeviceToHost, c->stream));
    if (dvar_out)
      HIPCHK(hipMemcpyAsync(dvar_out + (size_t)j * c->C * c->d, c->dvar.as<double>() + (size_t)j * c->pred_cap * c->d, row,
                            hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
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

static int finish_acq(bocf_ctx* c, double* acq_out) {
  c->have_acq = true;
  if (acq_out) HIPCHK(hipMemcpyAsync(acq_out, c->acq.p, sizeof(double) * c->C, hipMemcpyDeviceToHost, c->stream));
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

static int finish_acq_grad(bocf_ctx* c, double* acq_out, double* dacq_out) {
  if (dacq_out) HIPCHK(hipMemcpyAsync(dacq_out, c->dacq.p, sizeof(double) * (size_t)c->C * c->d, hipMemcpyDeviceToHost, c->stream));
  return finish_acq(c, acq_out);
}

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
      c->out_idx.ensure(sizeof(long long) * k) || c->out_val.ensure(sizeof(double) * k))
    return -1;
  {
    PhaseTimer t(c, "topk");
    launch_topk(c->acq.as<double>(), c->C, k, c->blk_idx.as<long long>(), c->blk_val.as<double>(), c->out_idx.as<long long>(),
                c->out_val.as<double>(), c->stream);
  }
  HIPCHK(hipMemcpyAsync(idx_out, c->out_idx.p, sizeof(long long) * k, hipMemcpyDeviceToHost, c->stream));
  if (val_out) HIPCHK(hipMemcpyAsync(val_out, c->out_val.p, sizeof(double) * k, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
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
