// Factorization schedules of libbocf_hip.so: the blocked Cholesky of all outputs -- resident workgroup teams (one launch, 2..8 panels;
// chol_team.hip), single stream with panel aggregation, reserved-CU chain with device-side counters -- and the triangular inverse by
// recursive doubling, early part underneath the factorization.  Called by bocf_fit (capi_fit.hip) through bocf_run_cholesky / bocf_run_trtri.
#include "bocf_ctx.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// (accessors of the factorization's buffers: one place for the casts)
static inline int wm(bocf_ctx* c) { return c->m; }
static inline double* wS(bocf_ctx* c) { return c->S.as<double>(); }
static inline double* wR(bocf_ctx* c) { return c->R.as<double>(); }
static inline double* wRT(bocf_ctx* c) { return c->RT.as<double>(); }
static inline double* wT(bocf_ctx* c) { return c->T.as<double>(); }
static inline double* wE(bocf_ctx* c) { return c->E.as<double>(); }
static inline double* wET(bocf_ctx* c) { return c->ET.as<double>(); }
static inline int* winfo(bocf_ctx* c) { return c->info.as<int>(); }

// ---------------------------------------------------------------------------------------------
// Cholesky (upper form, right-looking, NB = 128) of all m outputs at once.
//
// Per panel p the chain  diagonal block (one workgroup per output, ~41 us) -> row solve (one tile row) -> trailing
// update  is a dependency chain of short, latency-bound launches.  Schedules (option "lookahead"): 0 = everything on one stream, G panels
// per trailing update (option "aggregate"); 2 = the chain on reserved compute units with device-side counters (run_cholesky_reserved,
// one output, or two up to 12 panels).  (Round 3 built and measured three more -- a persistent chain on reserved CUs, output groups staggered
// on streams of their own, the next pair's diagonal block underneath the trailing update: slower, a tie, a tie -- and removed them again:
// DESIGN.md 10, profiles/r03.)
static GemmArgs trsm_args(bocf_ctx* c, int p, int W) {
  const int Np = c->Np;
  const long strideS = (long)Np * Np, strideE = (long)(Np / BOCF_TILE) * BOCF_TILE * BOCF_TILE;
  double* panel = wS(c) + (long)p * BOCF_TILE * Np + (long)(p + 1) * BOCF_TILE;
  GemmArgs g{};
  // U_p,> = E_p^T A_p,>   (in place)
  g.A = wE(c) + (long)p * BOCF_TILE * BOCF_TILE; g.lda = BOCF_TILE; g.strideA = strideE;
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
    double* panel = wS(c) + (long)p * BOCF_TILE * Np + (long)(p + 1) * BOCF_TILE;
    launch_tile128(wE(c) + (long)p * BOCF_TILE * BOCF_TILE, BOCF_TILE, strideE, panel, Np, strideS, panel, Np, strideS, 1.0, 0.0, wm(c), st,
                   W / BOCF_TILE);
  } else {
    launch_gemm_f64(trsm_args(c, p, W), wm(c), 0, st);
  }
}

// A_>,> -= U_p,>^T U_p,> restricted to block rows [first, first + rows) of the trailing matrix (tiles on/above the diagonal)
static GemmArgs syrk_args(bocf_ctx* c, int p, int first, int rows, int W) {
  const int Np = c->Np;
  const long strideS = (long)Np * Np;
  double* panel = wS(c) + (long)p * BOCF_TILE * Np + (long)(p + 1) * BOCF_TILE;
  const long off = (long)first * BOCF_TILE;
  GemmArgs t{};
  t.A = panel + off; t.lda = Np; t.strideA = strideS;
  t.B = panel + off; t.ldb = Np; t.strideB = strideS;
  double* trail = wS(c) + ((long)(p + 1) * BOCF_TILE + off) * Np + (long)(p + 1) * BOCF_TILE + off;
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
  for (hipStream_t* st : {&c->s_res, &c->s_hi, &c->s_bulk})
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
  if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&c->s_hi, (uint32_t)words, rest.data());
  if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&c->s_bulk, (uint32_t)words, rest.data());
  if (e != hipSuccess) {
    (void)hipGetLastError();
    for (hipStream_t* st : {&c->s_res, &c->s_hi, &c->s_bulk})
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
enum { MERGE_FIRST = 1, MERGE_SECOND = 2 };
static void merge_level(bocf_ctx* c, int lo, int w, int w2, int count, int which, hipStream_t st);

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
  double* S = wS(c);
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
  launch_potrf_diag(S, strideS, c->N, Np, 0, wE(c), wET(c), strideE, winfo(c), m, c->s_res, fP(0));
  for (int p = 0; p + 1 < nb; ++p) {
    const int W = Np - (p + 1) * BOCF_TILE;                // trailing width after panel p (>= 128)
    const int nrest = W / BOCF_TILE - 1;                   // tiles right of column block p+1
    double* panel = S + (long)p * BOCF_TILE * Np + (long)(p + 1) * BOCF_TILE;              // U[p][p+1 ...]
    double* trail = S + (long)(p + 1) * BOCF_TILE * Np + (long)(p + 1) * BOCF_TILE;        // A[p+1][p+1 ...]
    const double* Ep = wE(c) + (long)p * BOCF_TILE * BOCF_TILE;
    const int prev_rest = nrest + 1;                       // nrest of panel p-1
    // ---- chain: T1(p), S1(p), potrf(p+1)
    if (p > 0) launch_gate(fR(p - 1), 4 * prev_rest * m, fBA(p - 1), prev_rest * m, ferr, c->s_res);
    launch_tile128(Ep, BOCF_TILE, strideE, panel, Np, strideS, panel, Np, strideS, 1.0, 0.0, m, c->s_res, 1, BOCF_TILE, fT1(p));     // T1(p)
    launch_tile128(panel, Np, strideS, panel, Np, strideS, trail, Np, strideS, -1.0, 1.0, m, c->s_res, 1, BOCF_TILE, nullptr);      // S1(p)
    launch_potrf_diag(S, strideS, c->N, Np, p + 1, wE(c), wET(c), strideE, winfo(c), m, c->s_res, fP(p + 1));
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
  c->chol_err_off = 5 * nb;
#ifdef BOCF_PROBES
  if (getenv("BOCF_DBG")) {
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "run_cholesky_reserved: host enqueue %.1f us for %d panels\n", std::chrono::duration<double, std::micro>(t1 - t_host0).count(), nb);
  }
#else
  (void)t_host0;
#endif
  return 0;
}

// Called by the single-stream Cholesky schedules right after the row solve of panel p: once block rows [0, h) of U are final
// the part of the inverse that needs nothing else starts on the second stream, underneath the rest of the factorization
// (whose second half is a chain of short launches that leaves most of the chip idle).
static int maybe_start_early_inverse(bocf_ctx* c, int p) {
  const int nb = c->Np / BOCF_TILE;
  // (by size: from 24 panels -- N = 3072: 3.64 -> 3.57 ms, N = 3584: 4.84 -> 4.52; a tie below)
  const bool want = c->overlap_inverse > 0 || (c->overlap_inverse < 0 && nb >= 24 && (c->sched_m > 0 ? c->sched_m : c->m) >= 2);
  if (!want || nb < 8 || !c->s_inv) return 0;
  const int h = trtri_split(nb);
  if (!c->early_inverse_started && p == h - 1) {
    HIPCHK(hipEventRecord(c->ev_half, c->stream));
    HIPCHK(hipStreamWaitEvent(c->s_inv, c->ev_half, 0));
    trtri_early(c, h, c->s_inv);
    HIPCHK(hipEventRecord(c->ev_inv_early, c->s_inv));
    c->early_inverse_started = 1;
    // (a second stage -- the merges inside block rows [h, 3h/2) started when row 3h/2 - 1 is solved -- was measured in round 3: visible
    //  inverse 1.22 -> 1.21 ms, Cholesky 4.62 -> 4.69 ms at config 3: neutral, not kept)
  }
  return 0;
}

// A[pe ...][pe ...] -= U[p0 .. pe)[pe ...]^T U[p0 .. pe)[pe ...], pe = p0 + g: the trailing update behind a group of g solved block rows (K = 128 g)
static void launch_trailing_update(bocf_ctx* c, int p0, int g, hipStream_t st) {
  const int Np = c->Np, m = wm(c);
  const long strideS = (long)Np * Np;
  double* S = wS(c);
  const int pe = p0 + g;                            // first block row after the group
  const int W = Np - pe * BOCF_TILE;
  if (W <= 0) return;
  GemmArgs t{};
  double* rows = S + (long)p0 * BOCF_TILE * Np + (long)pe * BOCF_TILE;
  t.A = rows; t.lda = Np; t.strideA = strideS;
  t.B = rows; t.ldb = Np; t.strideB = strideS;
  double* trail = S + (long)pe * BOCF_TILE * Np + (long)pe * BOCF_TILE;
  t.Cin = trail; t.Cout = trail; t.ldc = Np; t.strideC = strideS;
  t.M = W; t.Ncols = W; t.K = g * BOCF_TILE; t.kb = g * BOCF_TILE; t.upper_only = 1; t.alpha = -1.0; t.beta = 1.0;
  // (epilogue 3 = the store epilogue on a grid of the upper tiles only.  A split-K treatment of the tiles of the last, partial round of a
  //  launch -- several workgroups per tile, partial sums combined by the tile's owner -- was built and measured in round 4: 4.19 ms against
  //  4.03 for this form at N = 4096, m = 4: tiles do not run in lockstep rounds, the tail packs by throughput; profiles/r04/split_tail.txt)
  launch_gemm_f64(t, m, 3, st);
}

// One panel group [p0, p0 + G) of the single-stream schedule for the current output window, on `st`.
// G panels per trailing update: the trailing matrix is read-modify-written once per G panels (its HBM traffic, not flops, is what the
// K = 128 updates cost); inside a group each new block row first receives the group's finished rows as ONE thin update with
// K = 128 * (rows so far).  G = 1: diagonal block, row solve, K = 128 trailing update.
static void chol_group_step(bocf_ctx* c, int p0, int G, hipStream_t st) {
  const int Np = c->Np, m = wm(c), nb = Np / BOCF_TILE;
  const long strideS = (long)Np * Np, strideE = (long)nb * BOCF_TILE * BOCF_TILE;
  double* S = wS(c);
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
        launch_tile128(rows, Np, strideS, rows, Np, strideS, row, Np, strideS, -1.0, 1.0, m, st, (W + BOCF_TILE) / BOCF_TILE, q * BOCF_TILE);
      else
        launch_gemm_f64(t, m, 0, st);
    }
    launch_potrf_diag(S, strideS, c->N, Np, p, wE(c), wET(c), strideE, winfo(c), m, st);
    launch_trsm(c, p, W, st);
  }
  launch_trailing_update(c, p0, g, st);
}


// Factorization AND inverse in one launch by resident workgroup teams (chol_team.hip), for models with few panels: the launched
// schedules below are then a chain of ~10 short dependent launches per panel with the chip idle underneath.  Returns 1 when the
// schedule does not apply (the caller falls through), 0 when it was enqueued, -1 on a HIP error.
static int run_cholesky_team(bocf_ctx* c, int G, int pfirst = 0, int pend = -1) {
  const int Np = c->Np, m = c->m, nb = Np / BOCF_TILE;
  if (nb < 2 || (G <= 0 && nb - pfirst > TEAM_MAX_NB)) return 1;
  if (c->ncu <= 0) {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    c->ncu = prop.multiProcessorCount;
  }
  const int ncu = c->force_cu_count > 0 ? c->force_cu_count : c->ncu;
  if (ncu < 4) return 1;
  if (pend < 0) pend = nb;
  const bool whole = (G <= 0 || G >= nb) && pend == nb;    // one launch: factorization and inverse (of the panels from pfirst on)
  const int g_max = whole ? nb - pfirst : G;
  // every workgroup of a launch must be resident at once: one 12-wave workgroup per compute unit at most
  const bool kinv = whole && pfirst == 0 && c->want_kinv;
  const int nt = nb - pfirst;
  const int units = whole ? 2 * (nt * (nt + 1) / 2 - 1) + nt * (nt - 1) + (kinv ? nb * (nb + 1) : 0) : 2 * (g_max * nb - 1);
  int mb = m < ncu / 2 ? m : ncu / 2;                      // outputs per launch
  int T = ncu / mb;
  if (pfirst > 0 && c->team_tail_share > 0) T = T * c->team_tail_share / 8;     // (hybrid: leave compute units to the inverse running underneath)
  if (T > 2 + units) T = 2 + units;                        // (the diagonal workgroup, the streaming workgroup, one per unit)
  if (T < 2) return 1;
  const long strideS = (long)Np * Np, strideE = (long)nb * BOCF_TILE * BOCF_TILE;
  const int words = chol_team_flag_words(nb);
  const size_t nflags = (size_t)m * words + 4;
  if (c->chol_flags.ensure(sizeof(int) * nflags)) return -1;
  int* F = c->chol_flags.as<int>();
  if (!c->flags_device_zeroed) HIPCHK(hipMemsetAsync(F, 0, sizeof(int) * nflags, c->stream));
#ifdef BOCF_PROBES
  const char* tl_path = getenv("BOCF_TEAM_TL");            // probes build: per-task stamps of every workgroup of the LAST launch (tools/team_timeline.py)
  const size_t tl_words = (size_t)ncu * 512 * 4;
  if (tl_path && !c->team_tl) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->team_tl), sizeof(unsigned long long) * tl_words));
#endif
  for (int p0 = pfirst; p0 < pend; p0 += g_max) {
    const int g = pend - p0 < g_max ? pend - p0 : g_max;
#ifdef BOCF_PROBES
    const char* tl_grp = getenv("BOCF_TEAM_TL_GROUP");     // which panel group's launch the timeline keeps (default: the last)
    const bool tl_this = tl_path && (!tl_grp || atoi(tl_grp) == p0 / g_max);
    if (tl_this) HIPCHK(hipMemsetAsync(c->team_tl, 0, sizeof(unsigned long long) * tl_words, c->stream));
#endif
    for (int j0 = 0; j0 < m; j0 += mb) {
      const int mr = m - j0 < mb ? m - j0 : mb;
      TeamArgs a{};
      a.S = wS(c) + (long)j0 * strideS; a.RT = wRT(c) + (long)j0 * strideS; a.R = wR(c) + (long)j0 * strideS; a.strideS = strideS;
      a.E = wE(c) + (long)j0 * strideE; a.ET = wET(c) + (long)j0 * strideE; a.strideE = strideE;
      a.N = c->N; a.Np = Np; a.nb = nb;
      a.info = winfo(c) + j0;
      a.F = F + (size_t)j0 * words; a.fstride = words;
      a.err = F + (size_t)m * words;
      a.T = T; a.p0 = p0; a.p1 = p0 + g; a.do_inverse = whole ? 1 : 0; a.tl = nullptr;
      a.KI = wT(c) + (long)j0 * strideS; a.do_kinv = kinv ? 1 : 0;
      // (streaming the critical tiles pays while the chain of diagonal blocks sets the pace: measured m = 4, 16 panels 1.11 -> 0.99 ms, 20: a tie, 24: 2.52 -> 2.63)
      a.crit_load = c->team_crit_load; a.stream = c->team_stream && g <= 20 ? 1 : 0;
#ifdef BOCF_PROBES
      a.tl = tl_this ? c->team_tl : nullptr;
#endif
      launch_chol_team(a, mr, c->stream);
    }
    if (!whole) {
      launch_trailing_update(c, p0, g, c->stream);
      for (int p = p0; p < p0 + g && pend == nb; ++p)      // (a caller that stops at pend starts the inverse itself)
        if (maybe_start_early_inverse(c, p)) return -1;
    }
  }
#ifdef BOCF_PROBES
  if (tl_path) {
    std::vector<unsigned long long> h(tl_words);
    HIPCHK(hipMemcpyAsync(h.data(), c->team_tl, sizeof(unsigned long long) * tl_words, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (FILE* f = fopen(tl_path, "w")) {
      fprintf(f, "# T %d nb %d m %d\n", T, nb, m);
      for (size_t b = 0; b < (size_t)ncu; ++b)
        for (size_t k = 0; k < 512; ++k) {
          const unsigned long long* r = h.data() + (b * 512 + k) * 4;
          if (r[0]) fprintf(f, "%zu %llu %llu %llu %llu\n", b, r[0], r[1], r[2], r[3]);
        }
      fclose(f);
    }
  }
#endif
  c->chol_flags_used = 1;
  c->chol_err_off = m * words;
  if (whole) {
    // (the teams write R^T (lower) and, transposed, R (upper): the strictly lower half of R / upper half of R^T is the zero half no fit ever writes)
    c->inverse_done = pfirst == 0 ? 1 : 0;
    c->kinv_done = kinv ? 1 : 0;
  }
  c->last_schedule = whole ? (pfirst == 0 ? 3 : 5) : 4;
  return 0;
}


// More than 24 panels: the launched schedule for the first h panels -- that is where the big trailing updates are, which the GEMM kernel does at
// 2-3x the rate of the teams' unit products -- and ONE team launch for the rest (Cholesky AND inverse of the trailing block: a problem of
// nb - h <= 24 panels, where the chain of diagonal blocks is what matters and the teams win).  h = the inverse's own split (largest power of
// two below nb): the inverse of the first h block rows and the first product of the top-level merge run on the second stream underneath,
// as in the launched schedule; after the team launch only R22 = (R22^T)^T and the merge's second product  R12 = -(R11 U12) R22  are left.
static int run_cholesky_hybrid(bocf_ctx* c, int G) {
  const int Np = c->Np, nb = Np / BOCF_TILE;
  const int h = trtri_split(nb);                           // (the split one level lower -- 8 + 24 panels at N = 4096 -- was measured: 6.5 ms against 4.9)
  if (nb - h > 24 || nb - h < 2 || !c->s_inv) return 1;
  if (c->ncu <= 0) {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    c->ncu = prop.multiProcessorCount;
  }
  if ((c->force_cu_count > 0 ? c->force_cu_count : c->ncu) < 4) return 1;
  if (c->team_hybrid == 2) {                               // (A/B: the first block rows by team launches of team_panels panels + trailing updates)
    const int rs0 = run_cholesky_team(c, c->team_panels > 0 ? c->team_panels : 4, 0, h);
    if (rs0 != 0) return rs0 < 0 ? -1 : 2;
  } else {
    for (int p0 = 0; p0 < h; p0 += G) chol_group_step(c, p0, h - p0 < G ? h - p0 : G, c->stream);
  }
  // the first h block rows of U are final: their inverse, and the first product of the top-level merge, underneath the team launch
  HIPCHK(hipEventRecord(c->ev_half, c->stream));
  HIPCHK(hipStreamWaitEvent(c->s_inv, c->ev_half, 0));
  trtri_early(c, h, c->s_inv);
  HIPCHK(hipEventRecord(c->ev_inv_early, c->s_inv));
  const int rs = run_cholesky_team(c, 0, h);
  if (rs != 0) return rs < 0 ? -1 : 2;                     // (2: the team launch does not apply after all -- the caller finishes with the launched schedule)
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_inv_early, 0));
  merge_level(c, 0, h, nb - h, 1, MERGE_SECOND, c->stream);
  c->inverse_done = 1;
  c->last_schedule = 5;
  return 0;
}

static int run_cholesky_impl(bocf_ctx* c);
int bocf_run_cholesky(bocf_ctx* c) {
#ifdef BOCF_PROBES
  const char* tl = getenv("BOCF_DBG_TL");                // plain-run timeline of the chain kernels (tools/dbg_timeline.py)
  if (tl) {
    HIPCHK(hipStreamSynchronize(c->stream));
    dbg_tl_start();
  }
  const int rc = run_cholesky_impl(c);
  if (tl && rc == 0) dbg_tl_dump(tl);
  return rc;
#else
  return run_cholesky_impl(c);
#endif
}
static int run_cholesky_impl(bocf_ctx* c) {
  const int Np = c->Np, m = c->m, nb = Np / BOCF_TILE;
  const long strideS = (long)Np * Np, strideE = (long)nb * BOCF_TILE * BOCF_TILE;
  double* S = wS(c);
  c->early_inverse_started = 0;
  c->inverse_done = 0;
  c->kinv_done = 0;
  set_potrf_scalar(c->potrf_scalar);                     // (probes build: timing-only variants of the diagonal-block kernel)
  // schedule: option "lookahead" = 2 (default by size: nb >= 8, at most 64 factorizations) -> reserved-CU lookahead
  // reserved-CU schedule with device-side dependencies: where the CHAIN of diagonal blocks sets the pace (few panels, or few
  // outputs per panel) it wins -- N = 2048 m = 4: 2.83 -> 2.52 ms, N = 3072: 5.4 -> 4.6, N = 4096 m = 1: 5.83 -> 4.57 -- where the
  // trailing updates do (N >= 6144 with m = 4: 17.7 vs 18.9 ms) the aggregated single-stream schedule below does.
  // "lookahead" = 2 forces it, -1 (default) chooses by size, 0 never uses it.  (Removed in round 3, all measured slower in plain runs and
  // kept until then for A/B: 1 = next panel's diagonal block + row solve on a second stream with stream events, 3 / 4 = panel pairs with
  // lookahead on two / three masked streams; their numbers are in DESIGN.md 10 and profiles/r02.)
  const int m_sched = c->sched_m > 0 ? c->sched_m : m;     // (a shard helper chooses as the replicated fit of ALL outputs would)
  // re-measured at the end of round 3 (tools/fit_schedule_sweep.sh, profiles/r03/fit_schedule_sweep.txt; the diagonal-block kernel, the row
  // products and the inverse all got faster since the rule was set, the cross-stream hand-overs did not): with two or more outputs the
  // single-stream schedule now wins from N = 2048 up by 10-50 % (N = 3072, m = 4: 3.62 against 5.38 ms; N = 4096, m = 2: 4.26 against 6.27);
  // the reserved-CU chain keeps ONE output (7-9 % at every size) and two outputs up to 12 panels (4 %)
  const bool reserved_auto = c->lookahead < 0 && nb >= 8 && ((m_sched == 1 && nb <= 32) || (m_sched == 2 && nb <= 12));
  // The gated (multi-stream) schedules are not used: after dependency time-outs (gated_off), for the redo of an attempt that timed out
  // (sched_retry), and for the FIRST factorization of a context -- it pays the one-time costs (code-object loads, allocations, stream
  // creation) that would otherwise sit between the launch of a polling kernel and the launch of the kernel it waits for.
  const bool gated_ok = c->cu_masks_ok && !c->gated_off && !c->sched_retry && c->fits_done > 0;
  // resident teams: few panels, not after dependency time-outs, not for the redo of an attempt that timed out
  const bool team_ok = !c->gated_off && !c->sched_retry && (c->team_fit > 0 || (c->lookahead < 0 && c->aggregate <= 0));
  // by size (m = 4, Cholesky + inverse in ms, launched / teams): 9 panels 1.05 / 0.56, 12: 1.34 / 0.75, 16: 1.76 / 1.10, 20: 2.56 / 1.8, 24: 3.32 / 2.51,
  // 32: 5.23 / 5.84 -- from there the K = 128 .. 512 unit products of the teams (~0.2 TFLOP/s per CU) lose to the launched GEMMs
  const int whole_max = c->team_whole_max;                 // up to here ONE team launch does everything; beyond, the hybrid schedule
  const bool team_auto = c->team_fit < 0 && nb >= 2 && nb <= 24;
  c->sched_retry = 0;
  if ((c->team_fit > 0 || team_auto) && team_ok && (nb <= whole_max || !c->team_hybrid)) {
    const int rs = run_cholesky_team(c, nb <= 24 ? 0 : c->team_panels);
    if (rs <= 0) return rs;
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
  // re-measured at the end of round 3 (m = 4, Cholesky ms for G = 1 / 2 / 3): N = 2048 1.27 / 1.19 / 1.20, 2560 1.77 / 1.69 / 1.65, 3072 3.05 / 2.92 / 3.06,
  // 3584 3.65 / 3.49 / 3.66, 4096 4.73 / 4.30 / 4.16, 5120 9.71 / 9.30 / 9.27, 6144 13.6 / 12.8 / 12.4: pairs from 16 panels, triples from 32 (with alpha
  // refined every G sits a decade inside the truth gate of tests/test_gpu_round3.py, so the choice is a matter of speed only)
  const int G_auto = nb >= 32 ? 3 : (nb >= 16 ? 2 : 1);
  const int G_use = c->aggregate > 0 ? c->aggregate : G_auto;
  // more than 24 panels: launched schedule for the first block rows, one team launch for the rest (run_cholesky_hybrid)
  if (team_ok && (c->team_fit > 0 || (c->team_fit < 0 && c->lookahead < 0 && c->aggregate <= 0)) && nb > whole_max && c->team_hybrid) {
    const int rs = run_cholesky_hybrid(c, G_use > 1 ? G_use : 1);
    if (rs <= 0) return rs;
  }
  if (G_use > 1 && nb >= 2 * G_use) {
    for (int p0 = 0; p0 < nb; p0 += G_use) {
      chol_group_step(c, p0, G_use, c->stream);
      for (int p = p0; p < p0 + G_use && p < nb; ++p)
        if (maybe_start_early_inverse(c, p)) return -1;
    }
    return 0;
  }
  for (int p = 0; p < nb; ++p) {
    chol_group_step(c, p, 1, c->stream);
    if (maybe_start_early_inverse(c, p)) return -1;
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
static void merge_level(bocf_ctx* c, int lo, int w, int w2, int count, int which, hipStream_t st) {
  const int Np = c->Np, m = wm(c);
  const long strideS = (long)Np * Np;
  const long dstep = (long)2 * w * BOCF_TILE * (Np + 1);            // next pair along the diagonal
  const long oLo = (long)lo * BOCF_TILE, oMid = (long)(lo + w) * BOCF_TILE;
  const int b1 = w * BOCF_TILE, b2 = w2 * BOCF_TILE;
  double* S = wS(c);
  double* R = wR(c);
  double* RT = wRT(c);
  double* T = wT(c);
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
  const int Np = c->Np, m = wm(c), nb = Np / BOCF_TILE;
  const long strideS = (long)Np * Np, strideE = (long)nb * BOCF_TILE * BOCF_TILE;
  launch_copy_diag_blocks(wE(c), strideE, wR(c), strideS, Np, blk_lo, blk_hi, m, st);
  launch_copy_diag_blocks(wET(c), strideE, wRT(c), strideS, Np, blk_lo, blk_hi, m, st);
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

// levels [level_lo, level_hi) of the whole inverse of the current window on `st` (level -1 = the copy of the inverted diagonal blocks,
// level l = the merges of width 2^l)
static void trtri_all(bocf_ctx* c, hipStream_t st, int level_lo, int level_hi) {
  const int nb = c->Np / BOCF_TILE;
  if (level_lo < 0) copy_diag_range(c, 0, nb, st);
  int lv = 0;
  for (int w = 1; w < nb; w *= 2, ++lv) {
    if (lv < level_lo || lv >= level_hi) continue;
    const int full = nb / (2 * w);                       // pairs with two complete halves
    if (full > 0) merge_level(c, 0, w, w, full, MERGE_FIRST | MERGE_SECOND, st);
    const int g = full * 2 * w;                          // a trailing incomplete pair, if any
    if (g + w < nb) merge_level(c, g, w, nb - (g + w), 1, MERGE_FIRST | MERGE_SECOND, st);
  }
}

int bocf_run_trtri(bocf_ctx* c, bool early_done) {
  if (c->inverse_done) return 0;                         // the team schedule produced R and R^T with the factorization
  const int nb = c->Np / BOCF_TILE;
  if (early_done) {                                      // the first h block rows were inverted underneath the factorization
    trtri_late(c, trtri_split(nb), c->stream);
    return 0;
  }
  // everything here: every level is ONE batched launch over all its pairs (the early / late split would double the
  // launch count, which is what the small sizes are made of)
  trtri_all(c, c->stream, -1, 64);
  return 0;
}

