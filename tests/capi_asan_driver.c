/* AddressSanitizer driver for the HOST side of the C ABI (include/bocf_hip.h), CPU only: every entry point is called the way a
 * careless binding would -- null context, null buffers, out-of-range sizes -- and must come back with a negative code and a
 * message, without touching memory it does not own (ASan) and without leaking (LeakSanitizer).  On a box without a GPU
 * bocf_create must fail cleanly too.  Built and run by tests/test_host_cpu.py::test_capi_asan_host_build. */
#include <stdio.h>
#include <string.h>
#include "../include/bocf_hip.h"

static int failures = 0;
#define EXPECT_NEG(call)                                                                         \
  do {                                                                                           \
    int rc_ = (call);                                                                            \
    const char* msg_ = bocf_last_error();                                                        \
    if (rc_ >= 0 || !msg_ || !msg_[0]) {                                                         \
      printf("FAIL %s -> %d (%s)\n", #call, rc_, msg_ ? msg_ : "(null)");                        \
      ++failures;                                                                                \
    }                                                                                            \
  } while (0)

/* f = sum (x - 0.25)^2 per row; user != NULL: abort on the second call */
static int quad_calls = 0;
static int quad_cb(void* user, const double* Z, const int* rows, int n, int d, double* f_out, double* g_out) {
  (void)rows;
  if (user && ++quad_calls >= 2) return 7;
  for (int i = 0; i < n; ++i) {
    double f = 0.0;
    for (int k = 0; k < d; ++k) {
      const double z = Z[i * d + k] - 0.25;
      f += z * z;
      g_out[i * d + k] = 2.0 * z;
    }
    f_out[i] = f;
  }
  return 0;
}

int main(void) {
  double buf[64];
  long long ibuf[16];
  int info[4];
  char id[BOCF_COMM_ID_BYTES];
  memset(buf, 0, sizeof(buf));
  if (bocf_version() <= 0) { printf("FAIL bocf_version\n"); ++failures; }
  EXPECT_NEG(bocf_create(0, NULL));
  bocf_ctx* ctx = NULL;
  int rc = bocf_create(1 << 20, &ctx);                 /* no such device (or no device at all) */
  if (rc >= 0 || ctx != NULL) { printf("FAIL bocf_create(bad device) -> %d\n", rc); ++failures; }
  bocf_destroy(NULL);
  EXPECT_NEG(bocf_set_option(NULL, "chunk", 128));
  EXPECT_NEG(bocf_sync(NULL));
  EXPECT_NEG(bocf_fit(NULL, buf, buf, 4, 2, 1, 0, buf, buf, buf, 5, buf, buf));
  EXPECT_NEG(bocf_infer(NULL, buf, buf, 4, 2, 1, 0, buf, buf, buf, 5, buf, buf, buf, buf, buf));
  EXPECT_NEG(bocf_update_targets(NULL, buf, buf));
  EXPECT_NEG(bocf_append(NULL, buf, buf, buf));
  EXPECT_NEG(bocf_lml_gradients(NULL, buf, buf, buf));
  EXPECT_NEG(bocf_last_fit_info(NULL, info, 1));
  EXPECT_NEG(bocf_get_factor(NULL, 0, buf, buf));
  EXPECT_NEG(bocf_get_train_kernel(NULL, 0, buf));
  EXPECT_NEG(bocf_set_posterior(NULL, 1, 1, 1, buf, buf, buf));
  EXPECT_NEG(bocf_set_candidates(NULL, buf, 4));
  EXPECT_NEG(bocf_predict(NULL, 0, buf, buf));
  EXPECT_NEG(bocf_predict_gradients(NULL, buf, buf));
  EXPECT_NEG(bocf_mean_at_train(NULL, buf));
  EXPECT_NEG(bocf_acq_linear(NULL, BOCF_ACQ_EI, buf, buf, 1, buf));
  EXPECT_NEG(bocf_acq_linear_grad(NULL, BOCF_ACQ_EI, buf, buf, 1, buf, buf));
  EXPECT_NEG(bocf_set_mc_samples(NULL, buf, 4));
  EXPECT_NEG(bocf_acq_mc(NULL, BOCF_ACQ_EI, BOCF_UTIL_LINEAR, buf, 0, buf, 1, buf, 1, buf));
  EXPECT_NEG(bocf_acq_mc_grad(NULL, BOCF_UTIL_LINEAR, buf, 0, buf, 1, buf, 1, buf, buf));
  EXPECT_NEG(bocf_select_topk(NULL, 4, ibuf, buf));
  EXPECT_NEG(bocf_global_topk(NULL, 4, 0, ibuf, buf));
  EXPECT_NEG(bocf_topk_packed(NULL, 4, 0, 1, 0, buf));
  EXPECT_NEG(bocf_merge_packed(NULL, 4, 1, buf, ibuf, buf));
  EXPECT_NEG(bocf_comm_unique_id(NULL));
  EXPECT_NEG(bocf_comm_init(NULL, id, 1, 0));
  EXPECT_NEG(bocf_comm_destroy(NULL));
  EXPECT_NEG(bocf_comm_info(NULL, info, info));
  EXPECT_NEG(bocf_profile_read(NULL, buf, ibuf, buf, 0));
  EXPECT_NEG(bocf_profile_phase(NULL, "kbuild", buf, ibuf, 0));
  EXPECT_NEG(bocf_hmc(NULL, buf, buf, 4, 2, 1, 0, buf, 2, info, 1.0, 0.5, buf, buf, 2, 2, 0.1, 5, 1, buf, info, info, info, ibuf));
  EXPECT_NEG(bocf_get_stat(NULL, "sched_timeouts", ibuf));
  EXPECT_NEG(bocf_set_kernel_ids(NULL, info, 1));
  EXPECT_NEG(bocf_predict_cov_column(NULL, 0, buf));
  /* the option table is host-only: enumerate it, check both ends of every range and one value outside (no GPU, no context) */
  {
    const int n = bocf_option_count();
    if (n < 20) { printf("FAIL bocf_option_count -> %d\n", n); ++failures; }
    for (int i = 0; i < n; ++i) {
      const char *name = NULL, *what = NULL;
      long long lo = 0, hi = 0;
      int kind = -1;
      if (bocf_option_info(i, &name, &lo, &hi, &kind, &what) != 0 || !name || !what || kind < 0 || kind > 1) {   /* kind 2 = probes: never in this library */
        printf("FAIL bocf_option_info(%d)\n", i);
        ++failures;
        continue;
      }
      if (bocf_option_check(name, lo) != 0 || bocf_option_check(name, hi) != 0) { printf("FAIL range ends of %s\n", name); ++failures; }
      EXPECT_NEG(bocf_option_check(name, hi + 1));
      EXPECT_NEG(bocf_option_check(name, lo - 1));
    }
    EXPECT_NEG(bocf_option_info(n, NULL, NULL, NULL, NULL, NULL));
    EXPECT_NEG(bocf_option_info(-1, NULL, NULL, NULL, NULL, NULL));
    EXPECT_NEG(bocf_option_check(NULL, 0));
    EXPECT_NEG(bocf_option_check("test_diag_shift_1e12", 1));
    EXPECT_NEG(bocf_option_check("kstar_valu_probe", 2));
  }
  {
    /* the batched L-BFGS-B (host arithmetic): argument checks, a whole run under the sanitizer, an aborting callback */
    double X0[6] = {0.9, 0.1, 0.25, 0.25, 0.0, 1.0}, lo[2] = {0.0, 0.0}, hi[2] = {1.0, 1.0}, X[6], F[3];
    long long calls[2] = {0, 0};
    int iters[3] = {0, 0, 0}, user = 1;
    if (bocf_lbfgsb_batched(NULL, NULL, X0, 3, 2, lo, hi, 100, 5, 1e6, 1e-8, 20, 1e-4, -1, X, F, calls, iters) != 1) { printf("FAIL lbfgsb(null callback)\n"); ++failures; }
    if (bocf_lbfgsb_batched(quad_cb, NULL, X0, 0, 2, lo, hi, 100, 5, 1e6, 1e-8, 20, 1e-4, -1, X, F, calls, iters) != 1) { printf("FAIL lbfgsb(A = 0)\n"); ++failures; }
    rc = bocf_lbfgsb_batched(quad_cb, NULL, X0, 3, 2, lo, hi, 100, 5, 1e6, 1e-8, 20, 1e-4, -1, X, F, calls, iters);
    if (rc != 0 || calls[0] < 2 || iters[1] != 0) { printf("FAIL lbfgsb run -> %d (%lld calls)\n", rc, calls[0]); ++failures; }
    for (int i = 0; i < 6; ++i)
      if (X[i] < 0.25 - 1e-5 || X[i] > 0.25 + 1e-5) { printf("FAIL lbfgsb optimum x[%d] = %g\n", i, X[i]); ++failures; }
    if (bocf_lbfgsb_batched(quad_cb, &user, X0, 3, 2, lo, hi, 100, 5, 1e6, 1e-8, 20, 1e-4, 3, X, F, NULL, NULL) != 2) { printf("FAIL lbfgsb(aborting callback)\n"); ++failures; }
  }
  if (failures) { printf("%d failure(s)\n", failures); return 1; }
  printf("capi asan driver: ok\n");
  return 0;
}
