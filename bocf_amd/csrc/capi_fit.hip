// Fit-side entry points of the C ABI (include/bocf_hip.h): data staging, alpha / refinement, bocf_fit (+ output-sharded form),
// bocf_update_targets, bocf_append, bocf_lml_gradients, bocf_infer, bocf_hmc and the inspection hooks.  The factorization schedules are
// capi_chol.hip, the context / options / predict / acquisition side is capi.hip.
#include "bocf_ctx.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// Per-output kernel families (bocf_set_kernel_ids): taken by the next fit / inference / chain whose output count matches (the list then
// overrides that call's scalar kernel_id for every output), cleared either way.
static int take_kernel_ids(bocf_ctx* c, int m) {
  c->kernel_ids.clear();
  const size_t pending = c->pending_ids.size();
  if ((int)pending == m && m > 0) {
    c->kernel_ids = c->pending_ids;
    c->kernel_id = c->kernel_ids[0];
  }
  c->pending_ids.clear();
  if (pending != 0 && (int)pending != m) return fail("bocf_set_kernel_ids", "the pending per-output kernel list has a different length than this call's outputs");
  return 0;
}
// (argument validation failed before the list could be taken: it must not survive into a later, unrelated call)
static int drop_kernel_ids(bocf_ctx* c, int rc) {
  if (c) c->pending_ids.clear();
  return rc;
}

extern "C" int bocf_set_kernel_ids(bocf_ctx* c, const int* ids, int m) {
  if (!c || (m > 0 && !ids)) return fail("bocf_set_kernel_ids", "null argument");
  if (m < 0 || m > BOCF_MAX_FITS) return fail("bocf_set_kernel_ids", "m out of range");
  for (int j = 0; j < m; ++j)
    if (ids[j] < 0 || ids[j] > 3) return fail("bocf_set_kernel_ids", "unknown kernel id");
  c->pending_ids.assign(ids, ids + m);
  return 0;
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
static int solve_alpha(bocf_ctx* c, bool refine_and_train_mean, bool with_lml = true) {
  const int N = c->N, Np = c->Np, m = c->m, nb = Np / BOCF_TILE;
  const long strideS = (long)Np * Np;
  launch_gemv_small_t(c->R.as<double>(), strideS, Np, c->yc.as<double>(), 1, Np, c->tvec.as<double>(), 1, m, c->stream);
  launch_gemv_upper_n(c->R.as<double>(), strideS, Np, c->tvec.as<double>(), c->alpha.as<double>(), m, c->stream);
  if (refine_and_train_mean) {
    if (c->meanpart.ensure(sizeof(double) * (size_t)2 * m * (Np / kalpha_block(Np)) * Np) || c->rvec.ensure(sizeof(double) * (size_t)m * Np) ||
        c->dvec.ensure(sizeof(double) * (size_t)m * Np) || c->mu_train.ensure(sizeof(double) * (size_t)m * Np))
      return -1;
    launch_kalpha_dd(c->Xs.as<double>(), c->xs_stride, N, Np, c->d, c->kernel_id, c->hypd.as<KernHyp>(), c->jit.as<double>(), c->alpha.as<double>(),
                     c->meanpart.as<double>(), m, c->stream, BOCF_KIDS(c));
    launch_refine_rhs(c->meanpart.as<double>(), N, Np, c->yc.as<double>(), c->rvec.as<double>(), m, c->stream);
    launch_gemv_small_t(c->R.as<double>(), strideS, Np, c->rvec.as<double>(), 1, Np, c->tvec.as<double>(), 1, m, c->stream);
    launch_gemv_upper_n(c->R.as<double>(), strideS, Np, c->tvec.as<double>(), c->dvec.as<double>(), m, c->stream);
    launch_refine_apply(c->dvec.as<double>(), N, Np, c->hypd.as<KernHyp>(), c->jit.as<double>(), c->yc.as<double>(), c->alpha.as<double>(),
                        c->mu_train.as<double>(), N, m, c->stream);
    c->mu_epoch++;
  }
  if (with_lml) launch_lml(c->S.as<double>(), strideS, N, Np, c->alpha.as<double>(), c->yc.as<double>(), c->lml.as<double>(), m, c->stream);
  return 0;
}

// X, the centred targets and the hyper-parameters onto the device (X, yc, hypd must be allocated).  With option
// "reuse_data" only the hyper-parameters move: X and Y are those of the previous call (same N, d, m).
// Host-to-device copies of a fit's small inputs through ONE pinned arena (a copy out of pageable memory is staged and waited for by the
// runtime, one after the other): the caller memcpy's into the slot and the copy is asynchronous.  The arena is free again at the fit's one
// stream synchronisation; slots of one fit do not overlap.  Returns nullptr when the arena cannot hold `bytes` at `off` (pageable path).
static char* fit_arena(bocf_ctx* c, size_t off, size_t bytes) {
  const size_t need = off + bytes;
  if (need > ((size_t)8 << 20)) return nullptr;
  if (need > c->up_pin_cap) {
    if (c->up_pin_cap && off > 0) return nullptr;          // (slots already handed out: do not move the arena under them)
    if (c->up_pin) (void)hipHostFree(c->up_pin);
    c->up_pin = nullptr;
    c->up_pin_cap = 0;
    size_t want = (size_t)1 << 16;
    while (want < need) want *= 2;
    if (hipHostMalloc(&c->up_pin, want, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      c->up_pin = nullptr;
      return nullptr;
    }
    c->up_pin_cap = want;
  }
  return static_cast<char*>(c->up_pin) + off;
}

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
    c->arena_used = 0;
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
    // one pinned arena: [X | hyper-parameters | centred targets | (jitter, in the ladder loop)] -- asynchronous copies, no synchronisation here
    const size_t bX = sizeof(double) * (size_t)N * d, bH = ((sizeof(KernHyp) * m + 63) / 64) * 64, bY = sizeof(double) * (size_t)m * Np;
    char* ar = fit_arena(c, 0, bX + bH + bY + 64 * (size_t)m);
    if (ar) {
      memcpy(ar, X, bX);
      memcpy(ar + bX, c->hyp.data(), sizeof(KernHyp) * m);
      memcpy(ar + bX + bH, yc.data(), bY);
      HIPCHK(hipMemcpyAsync(c->X.p, ar, bX, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(c->hypd.p, ar + bX, sizeof(KernHyp) * m, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(c->yc.p, ar + bX + bH, bY, hipMemcpyHostToDevice, c->stream));
      c->arena_used = bX + bH + bY;
    } else {
      HIPCHK(hipMemcpyAsync(c->X.p, X, bX, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(c->hypd.p, c->hyp.data(), sizeof(KernHyp) * m, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(c->yc.p, yc.data(), bY, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));   // host staging buffers go out of scope below
      c->arena_used = 0;
    }
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
    if (BOCF_KIDS(c)) hctx->pending_ids.assign(c->kernel_ids.begin() + j0, c->kernel_ids.begin() + j1);   // the share's kernel families
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
  const int simulate = c->shard_fit_simulate;                 // test hook (BOCF_PROBES builds): one process plays all G ranks in turn, no collectives
  const int G = simulate > 0 ? simulate : (c->comm ? c->world : 1), me = simulate > 0 ? 0 : (c->comm ? c->rank : 0);
  c->fitted = false; c->canned = false; c->have_acq = false; c->r32_valid = false; c->ri8_valid = false;
  const int Np = round_up(N, BOCF_TILE), nb = Np / BOCF_TILE;
  c->N = N; c->Np = Np; c->d = d; c->m = m; c->kernel_id = kernel_id;
  const int ids_rc = take_kernel_ids(c, m);                   // (a mismatch is reported from the local phase: nothing returns before the exchange)
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
    if (ids_rc) return -1;
    HIPCHK(hipSetDevice(c->device));
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
    hctx->overlap_inverse = c->overlap_inverse; hctx->potrf_scalar = c->potrf_scalar; hctx->team_fit = c->team_fit; hctx->team_panels = c->team_panels;
    hctx->team_crit_load = c->team_crit_load; hctx->team_stream = c->team_stream;
    hctx->trsm_wave = c->trsm_wave; hctx->merge_x3 = c->merge_x3; hctx->gated_off = c->gated_off;
    hctx->sched_m = m;
    // ... and with the caller's schedule HISTORY: whether CU masks work here, the pretended device size of the tests, and "not the first
    // factorization of the context" (the first one always runs single-stream), so that helper and replicated fit pick the same kernels
    hctx->cu_masks_ok = c->cu_masks_ok; hctx->force_cu_count = c->force_cu_count;
    if (hctx->fits_done < c->fits_done) hctx->fits_done = c->fits_done;
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
    c->mu_epoch++;
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
  if (!c || !X || !Y || !variance || !lengthscale || !noise) return drop_kernel_ids(c, fail("bocf_fit", "null argument"));
  if (N < 1 || d < 1 || d > BOCF_MAX_D || m < 1 || m > BOCF_MAX_FITS) return drop_kernel_ids(c, fail("bocf_fit", "N, d or m out of range"));
  if (kernel_id < 0 || kernel_id > 3) return drop_kernel_ids(c, fail("bocf_fit", "unknown kernel id"));
  for (int j = 0; j < m; ++j) {
    if (!(variance[j] > 0.0) || !(noise[j] >= 0.0)) return drop_kernel_ids(c, fail("bocf_fit", "variance must be > 0 and noise >= 0"));
    for (int q = 0; q < d; ++q)
      if (!(lengthscale[(long)j * d + q] > 0.0)) return drop_kernel_ids(c, fail("bocf_fit", "lengthscale must be > 0"));
  }
  if ((c->shard_fit && (c->comm || c->shard_fit_simulate > 0)) && !c->reuse_data && m > 1 && m % c->hyper_samples == 0)
    return fit_sharded(c, X, Y, N, d, m, kernel_id, variance, lengthscale, noise, max_jitter_tries, jitter_out, lml_out);
  HIPCHK(hipSetDevice(c->device));
  c->fitted = false;
  c->canned = false;
  c->sharded = false;
  c->have_acq = false;
  c->r32_valid = false; c->ri8_valid = false;
  const int Np = round_up(N, BOCF_TILE), nb = Np / BOCF_TILE;
  c->N = N; c->Np = Np; c->d = d; c->m = m; c->kernel_id = kernel_id;
  if (take_kernel_ids(c, m)) return -1;
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
    // (behind the fit's inputs in the pinned arena; attempt k + 1 rewrites the slot only after attempt k's synchronisation)
    char* jar = fit_arena(c, c->arena_used, sizeof(double) * m);
    if (jar) memcpy(jar, jeff.data(), sizeof(double) * m);
    HIPCHK(hipMemcpyAsync(c->jit.p, jar ? static_cast<const void*>(jar) : static_cast<const void*>(jeff.data()), sizeof(double) * m, hipMemcpyHostToDevice, c->stream));
    if (!jar) HIPCHK(hipStreamSynchronize(c->stream));     // (jeff goes out of scope)
    HIPCHK(hipMemsetAsync(c->info.p, 0, sizeof(int) * m, c->stream));
    {
      PhaseTimer t(c, "kbuild");
      launch_build_train_kernel(c->Xs.as<double>(), c->xs_stride, N, Np, d, kernel_id, c->hypd.as<KernHyp>(), c->jit.as<double>(), 1,
                                c->S.as<double>(), strideS, m, c->stream, BOCF_KIDS(c));
    }
    {
      PhaseTimer t(c, "cholesky");
      if (bocf_run_cholesky(c)) return -1;
      // (a failed attempt is rebuilt from scratch: the early inverse must be off the buffers first -- the wait costs nothing
      //  when the attempt succeeded, the inverse phase would wait for the same event)
      if (c->early_inverse_started) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_inv_early, 0));
    }
    // The inverse and alpha go out BEHIND the factorization before the host knows whether it succeeded (it almost always has: a failed
    // attempt only wastes them -- the diagonal-block kernel carries on with unit pivots, nothing reads out of bounds -- and is rebuilt from
    // scratch anyway): one synchronisation per fit instead of two, and no idle gap in front of the solves (75 us at N = 1024).
    {
      PhaseTimer t(c, "inverse");
      if (c->early_inverse_started) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_inv_early, 0));
      if (bocf_run_trtri(c, c->early_inverse_started != 0)) return -1;
    }
    {
      PhaseTimer t_alpha(c, "alpha");
      if (solve_alpha(c, !c->skip_mu_train)) return -1;
    }
    // status words and the log-marginal through one pinned block: [info (m ints) | schedule error word | pad] [log-marginal (m doubles)]
    const size_t ioff = ((sizeof(int) * (m + 1) + 7) / 8) * 8, pbytes = ioff + sizeof(double) * m;
    if (c->fit_pin_cap < pbytes) {
      if (c->fit_pin) (void)hipHostFree(c->fit_pin);
      c->fit_pin = nullptr; c->fit_pin_cap = 0;
      HIPCHK(hipHostMalloc(&c->fit_pin, pbytes < 4096 ? 4096 : pbytes, hipHostMallocDefault));
      c->fit_pin_cap = pbytes < 4096 ? 4096 : pbytes;
    }
    int* pin_i = static_cast<int*>(c->fit_pin);
    double* pin_l = reinterpret_cast<double*>(static_cast<char*>(c->fit_pin) + ioff);
    pin_i[m] = 0;
    HIPCHK(hipMemcpyAsync(pin_i, c->info.p, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
    if (c->chol_flags_used)
      HIPCHK(hipMemcpyAsync(pin_i + m, c->chol_flags.as<int>() + c->chol_err_off, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(pin_l, c->lml.p, sizeof(double) * m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int j = 0; j < m; ++j) info[j] = pin_i[j];
    int sched_err = pin_i[m];
#ifdef BOCF_PROBES
    if (c->force_sched_timeout && c->chol_flags_used) {      // test hook: as if a gate had run out of polls
      sched_err = 1;
      c->force_sched_timeout = 0;
    }
#endif
    c->chol_flags_used = 0;
#ifdef BOCF_PROBES
    if (sched_err && getenv("BOCF_DBG_FLAGS")) {             // which counters had arrived when the time-out fired
      std::vector<int> fl(c->chol_flags.cap / sizeof(int));
      (void)hipMemcpy(fl.data(), c->chol_flags.p, fl.size() * sizeof(int), hipMemcpyDeviceToHost);
      fprintf(stderr, "bocf_fit: dependency time-out (schedule %d, nb %d), first wait that ran out: id %d\n", c->last_schedule, nb, sched_err);
    }
#endif
    if (sched_err) {
      // A gate of a multi-stream schedule ran out of polls (0.2 s): its consumers ran on incomplete tiles.  That depends on timing
      // (a host stall while the streams are being filled, a tool that serialises dispatches across queues), not on the data:
      // rebuild K and redo THIS attempt on the single-stream schedule (c->sched_retry), count it; from the second time on the
      // gated schedules stay off for the context.
      c->sched_timeouts++;
      if (c->sched_timeouts >= 2) c->gated_off = 1;          // once may be a one-time stall (first use of a code object, a descheduled host thread); twice is a pattern
      if (c->sched_timeouts > 8) return drop_kernel_ids(c, fail("bocf_fit", "the factorization schedule keeps timing out waiting for device-side dependencies"));
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
    bocf_set_error("not positive definite, even with jitter.");
    return bad;
  }
  if (lml_out) memcpy(lml_out, reinterpret_cast<double*>(static_cast<char*>(c->fit_pin) + ((sizeof(int) * (m + 1) + 7) / 8) * 8), sizeof(double) * m);
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
                      c->meanpart.as<double>() + (size_t)m * nb * Np, 1, m, 1, c->stream, BOCF_KIDS(c));
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
  c->r32_valid = false; c->ri8_valid = false;
  if (c->mu_train.ensure(sizeof(double) * (size_t)m * c->N)) return -1;
  return refresh_targets(c, Y, lml_out);
}

// the launches of bocf_lml_gradients: Ky^-1 = R R^T (upper tiles, into the T scratch) and the hyper-gradient sums -> c->gout (m, 2 + d)
static int enqueue_lml_gradients(bocf_ctx* c, bool reduce = true) {
  const int N = c->N, Np = c->Np, m = c->m, d = c->d;
  const long strideS = (long)Np * Np;
  const int nblk = hypgrad_num_blocks(Np);
  DevBuf& part = c->gpart;
  DevBuf& out = c->gout;
  if (part.ensure(sizeof(double) * (size_t)m * nblk * (2 + d)) || out.ensure(sizeof(double) * (size_t)m * (2 + d))) return -1;
  // Kinv[r][c] = sum_{kk >= max(r,c)} RT[kk][r] RT[kk][c]   (unless the team schedule already accumulated it underneath the factorization)
  if (!c->kinv_done) {
  GemmArgs g{};
  g.A = c->RT.as<double>(); g.lda = Np; g.strideA = strideS;
  g.B = c->RT.as<double>(); g.ldb = Np; g.strideB = strideS;
  g.Cin = nullptr; g.Cout = c->T.as<double>(); g.ldc = Np; g.strideC = strideS;
  g.M = Np; g.Ncols = Np; g.K = Np; g.kb = Np; g.kbeg_ct = BOCF_TILE; g.upper_only = 1; g.alpha = 1.0;
  launch_gemm_f64(g, m, 3, c->stream);                     // (square, upper tiles only: no workgroups for the lower half)
  }
  launch_hypgrad(c->Xs.as<double>(), c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->alpha.as<double>(), c->T.as<double>(),
                 strideS, part.as<double>(), out.as<double>(), m, c->stream, BOCF_KIDS(c), reduce);
  return 0;
}

extern "C" int bocf_lml_gradients(bocf_ctx* c, double* dvariance_out, double* dlengthscale_out, double* dnoise_out) {
  if (!c || !c->fitted || c->canned) return fail("bocf_lml_gradients", "model not fitted");
  if (c->sharded) return fail("bocf_lml_gradients", "the fit is output-sharded (hyper-parameter learning factorizes unsharded)");
  HIPCHK(hipSetDevice(c->device));
  const int m = c->m, d = c->d;
  if (enqueue_lml_gradients(c)) return -1;
  std::vector<double> h((size_t)m * (2 + d));
  hipError_t e = hipMemcpyAsync(h.data(), c->gout.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream);
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
  if (!c || !X || !Y || !variance || !lengthscale || !noise) return drop_kernel_ids(c, fail("bocf_infer", "null argument"));
  const int Np = round_up(N < 1 ? 1 : N, BOCF_TILE);
  if (!c->fused_infer || Np != BOCF_TILE || d > BOCF_INFER_MAX_D) {
    const int sf = c->shard_fit;                         // an inference needs the upper factor on this rank: never output-sharded
    c->shard_fit = 0;
    c->want_kinv = 1;
    const int rc = bocf_fit(c, X, Y, N, d, m, kernel_id, variance, lengthscale, noise, max_jitter_tries, jitter_out, lml_out);
    c->shard_fit = sf;
    c->want_kinv = 0;
    if (rc) return rc;
    return bocf_lml_gradients(c, dvariance_out, dlengthscale_out, dnoise_out);
  }
  if (N < 1 || d < 1 || m < 1 || m > BOCF_MAX_FITS) return drop_kernel_ids(c, fail("bocf_infer", "N, d or m out of range"));
  if (kernel_id < 0 || kernel_id > 3) return drop_kernel_ids(c, fail("bocf_infer", "unknown kernel id"));
  for (int j = 0; j < m; ++j) {
    if (!(variance[j] > 0.0) || !(noise[j] >= 0.0)) return drop_kernel_ids(c, fail("bocf_infer", "variance must be > 0 and noise >= 0"));
    for (int q = 0; q < d; ++q)
      if (!(lengthscale[(long)j * d + q] > 0.0)) return drop_kernel_ids(c, fail("bocf_infer", "lengthscale must be > 0"));
  }
  HIPCHK(hipSetDevice(c->device));
  c->fitted = false;
  c->canned = false;
  c->have_acq = false;
  c->r32_valid = false; c->ri8_valid = false;
  c->N = N; c->Np = Np; c->d = d; c->m = m; c->kernel_id = kernel_id;
  if (take_kernel_ids(c, m)) return -1;
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
    launch_infer128(c->X.as<double>(), N, d, kernel_id, c->hypd.as<KernHyp>(), c->yc.as<double>(), out_dev, m, c->stream, BOCF_KIDS(c));
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
    bocf_set_error("not positive definite, even with jitter.");
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
    return drop_kernel_ids(c, fail("bocf_hmc", "null argument"));
  if (N < 1 || N > BOCF_TILE || d < 1 || d > BOCF_INFER_MAX_D || m < 1 || m > BOCF_MAX_FITS) return drop_kernel_ids(c, fail("bocf_hmc", "N (<= 128), d (<= 16) or m out of range"));
  if (kernel_id < 0 || kernel_id > 3) return drop_kernel_ids(c, fail("bocf_hmc", "unknown kernel id"));
  if (nls != 1 && nls != d) return drop_kernel_ids(c, fail("bocf_hmc", "nls must be 1 (isotropic) or d (ARD)"));
  if (num_samples < 1 || hmc_iters < 1 || !(stepsize > 0.0) || !(prior_a > 0.0) || !(prior_b > 0.0) || max_jitter_tries < 0)
    return drop_kernel_ids(c, fail("bocf_hmc", "num_samples, hmc_iters, stepsize, prior or max_jitter_tries out of range"));
  const int P = 2 + nls, Np = BOCF_TILE;
  for (int j = 0; j < m; ++j) {
    int nfree = 0;
    for (int k = 0; k < P; ++k) {
      const double t = theta[(size_t)j * P + k];
      if (!(t > 0.0) && !(k == P - 1 && t == 0.0)) return drop_kernel_ids(c, fail("bocf_hmc", "theta must be positive (noise >= 0)"));
      nfree += fixed[(size_t)j * P + k] ? 0 : 1;
    }
    if (nfree < 1) return drop_kernel_ids(c, fail("bocf_hmc", "an output has no free parameter"));
  }
  HIPCHK(hipSetDevice(c->device));
  c->fitted = false; c->canned = false; c->have_acq = false; c->r32_valid = false; c->ri8_valid = false;
  c->N = N; c->Np = Np; c->d = d; c->m = m; c->kernel_id = kernel_id;
  if (take_kernel_ids(c, m)) return -1;
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
  launch_hmc128(a, kernel_id, m, c->stream, BOCF_KIDS(c));
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
  if (bad) bocf_set_error("not positive definite, even with jitter.");
  return bad;
}

// The device work of ONE inference at the hyper-parameters in c->hypd, enqueued on the context's stream without any host interaction:
// what bocf_fit (attempt 0 of the ladder, no refinement / train mean) + bocf_lml_gradients launch.  Results: c->lml, c->gout, c->info.
static int enqueue_inference(bocf_ctx* c) {
  const int N = c->N, Np = c->Np, m = c->m, d = c->d;
  const long strideS = (long)Np * Np;
  // (the inputs were scaled, the schedule counters zeroed, by the PRE launch of hmc_stream_kernel; its POST launch takes the log-marginal
  //  and reduces the gradient partials)
  launch_build_train_kernel(c->Xs.as<double>(), c->xs_stride, N, Np, d, c->kernel_id, c->hypd.as<KernHyp>(), c->jit.as<double>(), 1,
                            c->S.as<double>(), strideS, m, c->stream, BOCF_KIDS(c));
  if (bocf_run_cholesky(c)) return -1;
  if (c->early_inverse_started) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_inv_early, 0));
  if (bocf_run_trtri(c, c->early_inverse_started != 0)) return -1;
  if (solve_alpha(c, false, false)) return -1;
  return enqueue_lml_gradients(c, false);
}

// The HMC chain of GPModel.updateModel for models beyond the fused chain (N > 128 or d > 16), STREAM-RESIDENT: every leapfrog step is
// the launch sequence of one inference between two launches of hmc_stream_kernel (hmc_stream.hip), all on the context's stream; the host
// only enqueues, and looks at the abort word every few draws.  Arguments as bocf_hmc.  The chain stops early -- *draws_done_out < num_samples,
// every output back at the start of that draw -- when a factorization meets a non-positive pivot (jitchol's ladder, linalg.py:52-71, is the
// host's: the caller runs that draw with bocf_infer per step and calls again for the rest), when parameters leave the positive domain,
// or when a factorization schedule timed out.  accepted / diverged count the completed draws only.
extern "C" int bocf_hmc_streamed(bocf_ctx* c, const double* X, const double* Y, int N, int d, int m, int kernel_id, double* theta, int nls,
                                 const int* fixed, double prior_a, double prior_b, const double* momenta, const double* uniforms, int num_samples,
                                 int hmc_iters, double stepsize, double* chains_out, int* accepted_out, int* diverged_out, int* draws_done_out,
                                 long long* inferences_out) {
  if (!c || !X || !Y || !theta || !fixed || !momenta || !uniforms || !chains_out || !accepted_out || !draws_done_out)
    return drop_kernel_ids(c, fail("bocf_hmc_streamed", "null argument"));
  if (N < 1 || d < 1 || d > BOCF_MAX_D || m < 1 || m > BOCF_MAX_FITS) return drop_kernel_ids(c, fail("bocf_hmc_streamed", "N, d or m out of range"));
  if (kernel_id < 0 || kernel_id > 3) return drop_kernel_ids(c, fail("bocf_hmc_streamed", "unknown kernel id"));
  if (nls != 1 && nls != d) return drop_kernel_ids(c, fail("bocf_hmc_streamed", "nls must be 1 (isotropic) or d (ARD)"));
  if (num_samples < 1 || hmc_iters < 1 || !(stepsize > 0.0) || !(prior_a > 0.0) || !(prior_b > 0.0))
    return drop_kernel_ids(c, fail("bocf_hmc_streamed", "num_samples, hmc_iters, stepsize or prior out of range"));
  const int P = 2 + nls;
  for (int j = 0; j < m; ++j) {
    int nfree = 0;
    for (int k = 0; k < P; ++k) {
      const double t = theta[(size_t)j * P + k];
      if (!(t > 0.0) && !(k == P - 1 && t == 0.0)) return drop_kernel_ids(c, fail("bocf_hmc_streamed", "theta must be positive (noise >= 0)"));
      nfree += fixed[(size_t)j * P + k] ? 0 : 1;
    }
    if (nfree < 1) return drop_kernel_ids(c, fail("bocf_hmc_streamed", "an output has no free parameter"));
  }
  *draws_done_out = 0;
  if (inferences_out) *inferences_out = 0;
  struct Unset { bocf_ctx* c; ~Unset() { c->flags_device_zeroed = 0; c->want_kinv = 0; } } unset{c};
  // ---- one ordinary inference at the starting point: stages X / Y (unless option reuse_data says they are resident), sizes every buffer,
  //      pays the one-time costs.  A start that needs jitter is the host's (draws_done = 0).
  {
    std::vector<double> var(m), ls((size_t)m * d), nz(m);
    for (int j = 0; j < m; ++j) {
      var[j] = theta[(size_t)j * P];
      nz[j] = theta[(size_t)j * P + P - 1];
      for (int q = 0; q < d; ++q) ls[(size_t)j * d + q] = theta[(size_t)j * P + 1 + (nls == 1 ? 0 : q)];
    }
    const int sf = c->shard_fit, smt = c->skip_mu_train;
    c->shard_fit = 0; c->skip_mu_train = 1; c->want_kinv = 1;
    int rc = bocf_fit(c, X, Y, N, d, m, kernel_id, var.data(), ls.data(), nz.data(), 0, nullptr, nullptr);
    if (rc == 0) rc = bocf_lml_gradients(c, nullptr, nullptr, nullptr);
    c->shard_fit = sf; c->skip_mu_train = smt;
    if (rc < 0) return -1;
    if (rc > 0) return 0;
  }
  HIPCHK(hipSetDevice(c->device));
  c->have_acq = false;
  const size_t nth = (size_t)m * P, nmom = (size_t)m * num_samples * P, nuni = (size_t)m * num_samples;
  const size_t nstate = (size_t)hmc_stream_state_doubles(m);
  // one scratch block: theta | momenta | uniforms | chains | state (doubles), n_eval (long long), fixed | accepted | diverged | abort (ints)
  const size_t dbl = nth + nmom + nuni + nmom + nstate, ints = nth + 2 * (size_t)m + 4;
  if (c->hmc_buf.ensure(sizeof(double) * dbl + sizeof(long long) * m + sizeof(int) * ints)) return -1;
  double* dth = c->hmc_buf.as<double>();
  double* dmom = dth + nth;
  double* duni = dmom + nmom;
  double* dch = duni + nuni;
  double* dstate = dch + nmom;
  long long* dninf = reinterpret_cast<long long*>(dstate + nstate);
  int* dfix = reinterpret_cast<int*>(dninf + m);
  int* dacc = dfix + nth;
  int* ddiv = dacc + m;
  int* dabort = ddiv + m;
  const int minus1 = -1;
  HIPCHK(hipMemcpyAsync(dth, theta, sizeof(double) * nth, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dmom, momenta, sizeof(double) * nmom, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(duni, uniforms, sizeof(double) * nuni, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dfix, fixed, sizeof(int) * nth, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(dch, 0, sizeof(double) * (nmom + nstate), c->stream));
  HIPCHK(hipMemsetAsync(dninf, 0, sizeof(long long) * m, c->stream));
  HIPCHK(hipMemsetAsync(dacc, 0, sizeof(int) * 2 * (size_t)m, c->stream));
  HIPCHK(hipMemcpyAsync(dabort, &minus1, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HmcStreamArgs a{};
  a.m = m; a.P = P; a.nls = nls; a.d = d; a.ns = num_samples; a.iters = hmc_iters;
  a.eps = stepsize; a.prior_a = prior_a; a.prior_b = prior_b; a.prior_const = -lgamma(prior_a) + prior_a * log(prior_b);   // priors.py:271
  a.diag_shift = c->test_diag_shift;
  a.fixed = dfix; a.mom = dmom; a.uni = duni; a.state = dstate; a.chains = dch; a.accepted = dacc; a.diverged = ddiv; a.n_eval = dninf;
  a.abort_draw = dabort; a.hyp = c->hypd.as<KernHyp>();
  a.info = c->info.as<int>(); a.theta = dth;
  a.X = c->X.as<double>(); a.Xs = c->Xs.as<double>(); a.strideXs = c->xs_stride; a.N = N; a.Np = c->Np;
  a.S = c->S.as<double>(); a.strideS = (long)c->Np * c->Np; a.alpha = c->alpha.as<double>(); a.yc = c->yc.as<double>();
  a.part = c->gpart.as<double>(); a.nblk = hypgrad_num_blocks(c->Np);
  // the schedule of the fit above is the schedule of every factorization of the chain (same shape, same options); when it is the team
  // schedule its counters are zeroed by the PRE launches instead of a memset node per step
  a.flags = nullptr; a.flag_words = 0; a.sched_err = nullptr;
  if (c->last_schedule == 3) {
    a.flags = c->chol_flags.as<int>(); a.flag_words = c->chol_err_off / m; a.sched_err = a.flags + c->chol_err_off;
    c->flags_device_zeroed = 1;
  } else if (c->last_schedule == 2 || c->last_schedule == 4 || c->last_schedule == 5) {
    a.sched_err = c->chol_flags.as<int>() + c->chol_err_off;
  }
  launch_hmc_stream(a, HS_INIT, 0, 0, c->stream);
  if (enqueue_inference(c)) return -1;
  launch_hmc_stream(a, HS_EVAL0, 0, 0, c->stream);
  const int check_every = 8;                               // draws between two looks at the abort word (a sync each)
  int aborted = -1;
  for (int i = 0; i < num_samples && aborted < 0; ++i) {
    for (int it = 0; it < hmc_iters; ++it) {
      launch_hmc_stream(a, HS_PRE, i, it, c->stream);
      if (enqueue_inference(c)) return -1;
      launch_hmc_stream(a, HS_POST, i, it, c->stream);
    }
    if ((i + 1) % check_every == 0 || i + 1 == num_samples) {
      HIPCHK(hipMemcpyAsync(&aborted, dabort, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
    }
  }
  launch_hmc_stream(a, HS_FINISH, 0, 0, c->stream);
  std::vector<long long> ninf(m, 0);
  std::vector<int> dv(m, 0);
  HIPCHK(hipMemcpyAsync(theta, dth, sizeof(double) * nth, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(chains_out, dch, sizeof(double) * nmom, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(accepted_out, dacc, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(dv.data(), ddiv, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(ninf.data(), dninf, sizeof(long long) * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  LAUNCHCHK();
  c->chol_flags_used = 0;
  c->fitted = false;                                       // the factor on the device belongs to the chain's last trajectory point
  if (diverged_out) memcpy(diverged_out, dv.data(), sizeof(int) * m);
  long long total = 0;
  for (int j = 0; j < m; ++j) total = ninf[j] > total ? ninf[j] : total;
  if (inferences_out) *inferences_out = total;
  *draws_done_out = aborted >= 0 ? aborted : num_samples;
  return 0;
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
  launch_build_train_kernel(c->Xs.as<double>() + (long)j * c->xs_stride, 0, N, Np, c->d, BOCF_KIDS(c) ? c->kernel_ids[j] : c->kernel_id, c->hypd.as<KernHyp>() + j, nullptr, 0,
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

