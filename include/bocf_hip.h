/*
 * bocf_hip.h -- C ABI of libbocf_hip.so: MI355X (gfx950) implementation of BOCF's
 * GP-posterior + composite-acquisition hot path.
 *
 * The reference (RaulAstudillo06/BOCF) has NO native boundary on this path: it is pure
 * Python over SciPy LAPACK.  This header is the FFI a maintainer binds with ctypes
 * underneath the reference's two plug-in surfaces (multi_outputGP.py:9-349 and
 * GPyOpt/acquisitions/base.py:5-74); INTEGRATION.md shows that binding.  Every entry
 * point names the reference code it replaces (paths relative to the reference root).
 *
 * Conventions: plain C, caller-owned HOST buffers unless a name says "device", row-major
 * float64, no alignment requirement.  Return 0 = ok; >0 = LAPACK-style info (1-based index
 * of the first non-positive Cholesky pivot); <0 = HIP/runtime/argument error (text via
 * bocf_last_error()).  A context owns its device memory and one HIP stream, is bound to one
 * GPU, is NOT thread-safe, and every call is synchronous on return unless stated otherwise.
 */
#ifndef BOCF_HIP_H
#define BOCF_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bocf_ctx bocf_ctx;

/* kernel ids: GPy/kern/src/rbf.py:42-43 (RBF) and GPy/kern/src/se.py:44-62 (SE, the GPModel
 * default, gpmodel.py:58) share sigma^2 exp(-r^2/2); Matern52 stationary.py:529-530;
 * Matern32 stationary.py:440-441.  Distances use direct differences of the inputs divided
 * by the lengthscale (se.py:65-93 convention). */
enum { BOCF_KERN_RBF = 0, BOCF_KERN_SE = 1, BOCF_KERN_MATERN52 = 2, BOCF_KERN_MATERN32 = 3 };

/* predict flags */
enum { BOCF_ADD_NOISE = 1,   /* + sigma_n^2: GP.predict gp.py:316-320 / posterior_variance gp.py:416-417 */
       BOCF_CLIP = 2 };      /* clip variance to [1e-10, inf): GPyOpt/models/gpmodel.py:147,174,183 */

/* acquisition kinds */
enum { BOCF_ACQ_EI = 0,      /* maEI.py:96 / uEI_noiseless.py:80 */
       BOCF_ACQ_PI = 1 };    /* maPI.py:91-92 / uPI.py:83 (jitter 1e-6) */

/* device utility functions U(theta, y), y in R^m -- the closed set the reference's experiment
 * scripts use (arbitrary Python callables cannot run on the device):
 *   LINEAR       theta . y                              test_1b.py:89-90     theta_dim = m
 *   NEG_SQ_DIST  -sum_j (y_j - theta_j)^2               test_1a.py:89-92     theta_dim = m
 *   NEG_SUM_EXP  -sum_j exp(y_j)                        test_2a.py:60-62     theta unused
 *   NEG_EXP_COS  -sum_j c_j e^{-y_j/pi} cos(pi y_j)     test_3a.py:52-57     util_params = c (m)
 *   ROSENBROCK   -sum_{j<m/2} (a-y_j)^2 + 100 y_{j+m/2}^2   test_5a.py:48-52  a = theta[0] */
enum { BOCF_UTIL_LINEAR = 0, BOCF_UTIL_NEG_SQ_DIST = 1, BOCF_UTIL_NEG_SUM_EXP = 2,
       BOCF_UTIL_NEG_EXP_COS = 3, BOCF_UTIL_ROSENBROCK = 4 };

int bocf_version(void);
const char* bocf_last_error(void);

/* Create / destroy a context on HIP device `device`. */
int bocf_create(int device, bocf_ctx** out);
void bocf_destroy(bocf_ctx* ctx);

/* Options (bocf_option_info enumerates the table: name, range, kind; kind 0 = speed only, kind 1 = semantics the caller asks for).
 * Sizes / plumbing:
 *   "chunk" = max candidates per pass (multiple of 128; default 65536), "workspace_mb" = cap of the per-pass K* workspace (default
 *   24576; the chunk is lowered to fit), "profile" = 1 records HIP events around the dominant (variance-GEMM) kernel and the named phases,
 *   "small_path" = 0 disables the GEMV-shaped path for <= 16 candidates, "overlap" = 1 builds K* on a second stream, "prefetch1".
 * Semantics (kind 1):
 *   "predict_f32" = 1 runs the O(N^2 C) variance contraction in fp32 (K* and the inverse factor rounded to fp32, fp32 MFMA; fit, mean and
 *     gradients stay fp64) -- the arithmetic BASELINE configs[4] names,
 *   "reuse_data" = 1: the following bocf_fit calls use the X / Y of the previous fit (same N, d, m; the pointers are ignored) and upload
 *     only the hyper-parameters; "skip_mu_train" = 1: bocf_fit does not refresh the posterior mean at the training inputs -- both for
 *     the thousands of inferences of a hyper-parameter update (optimise + HMC), which read only the log-marginal and its gradients;
 *     reset both to 0 before the fit that serves predictions,
 *   "hyper_samples" = H (default 1): the m outputs given to bocf_fit are H hyper-samples x m/H model outputs, hyper-sample-major -- the
 *     model_instances of GPModel (gpmodel.py:80-96, one kernel/noise setting per HMC draw).  The acquisition entry points then run the
 *     reference's h-loop (maEI.py:85-97, uEI_noiseless.py:71-82) on the device: theta/W refer to the m/H model outputs, every
 *     hyper-sample adds 1/H of its marginal,
 *   "acq_hyper_samples" = n (default 0 = all): the acquisitions average over the first n hyper-samples only (n_hyps_samples =
 *     min(10, number_of_hyps_samples()), maEI.py:35),
 *   "best_group" = -1 (default): hyper-sample h uses its own best-so-far (maEI.py:88); g >= 0: every h uses hyper-sample g's
 *     (uEI_noiseless.py:66 evaluates it once, with whichever hyper-sample was current).
 * Factorization schedules (speed only: every one computes the same factor up to rounding; tests/test_gpu_parity.py pins it):
 *   "team_fit" = -1 (default: by size) / 0 (never) / 1: resident workgroup TEAMS that hand tiles to each other through device-side
 *     counters (chol_team.hip).  Up to "team_whole_max" (default 24) panels of 128 rows ONE launch does the Cholesky AND the inverse;
 *     beyond, "team_hybrid" = 2 (default): the first block rows as team launches of "team_panels" (default 6) panels, each followed by one
 *     trailing update with K = 128 x that, then ONE team launch (Cholesky + inverse) for the remaining corner on "team_tail_share"
 *     eighths of the compute units while the inverse of the first block rows runs underneath; 1: the first block rows by the launched
 *     schedule; 0: no hybrid (team_fit = 1 then means panel groups throughout).  "team_stream" = 1 (default): a workgroup of every team
 *     forms U[p][p+1] and the last row of A[p+1][p+1] sixteen rows at a time UNDERNEATH the diagonal block p (same sums in the same
 *     order: bit-identical to 0 up to 8 panels); "team_crit_load": the workgroups of the critical tiles carry nothing else while the
 *     others get by with at most this many tiles each,
 *   "predict_i8" = 1: the variance contraction v = L^-1 K*, sum v^2 (posterior.py:308-313) in EXACT int8 products (gemm_i8.hip): every
 *     column of R and of K* gets one power-of-two scale and is cut into six signed digits of radix 254 (|d| <= 127: 47.9 bits), the 21 digit
 *     products with i + j <= 5 run on v_mfma_i32_16x16x64_i8 into int32 sums, fp64 recombination, squares and the per-128-row partial sums of
 *     the fp64 kernels.  From 17 candidates per call up, N <= 16384, variances only (gradients keep the fp64 path); the posterior mean does not
 *     go through it.  Accuracy at cond(Ky) = 4e9 (config 3): |d var| <= 1.1e-11 sigma_f^2, 3.4e-6 relative at variances of 1e-6 sigma_f^2 (the fp64
 *     contraction: 1e-14 / 1e-8) -- inside every parity gate of tests/test_gpu_round3.py.  Speed at N = 4096: 38-40 against 62.2 ms per
 *     65 536-candidate step.  An option like predict_f32: the default stays the fp64 contraction.  "i8_group" (speed only): 0 = the
 *     workgroups an XCD runs together are a block of 4 row-tile pairs x 8 column tiles, g >= 1 = bands of g pairs x all column tiles.
 *   "lookahead" = -1 (by size) / 0 (single stream) / 2 (reserved-CU chain with device-side counters: one output, or two outputs up to
 *     12 panels), "aggregate" = G panels per trailing update of the single-stream schedule (default 0 = by size: 1 below 16 panels,
 *     2 from 16, 3 from 32), "overlap_inverse" (early part of the inverse underneath the factorization), "trsm_wave", "merge_x3" = 0 /
 *     1 / 2 (second product of an inverse merge in the three-buffer triangular kernel: never / from 4096 rows / whenever the shape allows).
 *     Every schedule that waits on device-side counters has a 0.2 s cut-off; when it fires the attempt is redone on the single-stream
 *     schedule (bocf_get_stat "sched_timeouts"), twice and those schedules stay off for the context.
 *   "swizzle" = variance-GEMM tiling: -1 (default) by size (258 from 2048 candidates per pass when the padded N is a multiple of 256,
 *     else 0); 258 = 256-row tiles in the three-buffer kernel, 0 = 128-row tiles; both give bit-identical sums.
 * Multi-GPU:
 *   "shard_fit" = 1 (with a communicator, bocf_comm_init): bocf_fit factorizes only this rank's contiguous share of the m independent
 *     outputs (multi_outputGP.py:64-102 fits them one after the other) and the ranks exchange what prediction needs -- the inverse
 *     factors by one RCCL broadcast per output (one group), alpha / train mean / log-marginal / jitter / status by ONE all-reduce.  The
 *     helper context takes the caller's schedule options and history and chooses by the global output count, so a share is factorized
 *     by the kernel sequence the replicated fit would run for it (bit-identical results); where the schedules still differ the results
 *     agree up to rounding.  bocf_get_factor (L), bocf_append, bocf_update_targets and bocf_lml_gradients are not served by such a
 *     fit (they need the upper factor, which stays on its owner).
 * NOT in this library: the timing-only kernel variants, slower tilings / tile orders kept for the tools, and the test hooks (diagonal
 * shift that forces the jitter ladder, single-process stand-in for the ranks of a sharded fit, forced schedule time-out / device size)
 * exist only in libbocf_hip_probes.so (the same sources with -DBOCF_PROBES), which tools/ and the tests that need a hook load; the
 * product library rejects those names. */
int bocf_set_option(bocf_ctx* ctx, const char* name, long long value);

/* FIT, one exact-GP inference per output with fixed hyper-parameters.  Replaces
 * multi_outputGP.updateModel (multi_outputGP.py:97-102) -> GPModelFixedHyps.updateModel
 * (gpmodel_fixed_hyps.py:61-69) -> GP.parameters_changed (GPy/core/gp.py:247-256) ->
 * ExactGaussianInference.inference (exact_gaussian_inference.py:29-65):
 *   Y is mean-centred per output (Standardize, normalizer.py:57-63, std == 1);
 *   Ky = K(X,X) + (noise + 1e-8) I (:46-47); L = chol(Ky) with the jitchol ladder
 *   (GPy/util/linalg.py:52-71: jitter = mean(diag) 1e-6, x10 per retry, <= max_tries);
 *   alpha = Ky^-1 (Y - mean) (:51); log-marginal (:53).
 * X (N,d); Y (m,N); variance (m); lengthscale (m,d) (an isotropic kernel repeats its value);
 * noise (m).  jitter_out (m) / lml_out (m) may be NULL.  Returns 0, or info>0 if some output
 * is not positive definite even with jitter (scipy.linalg.LinAlgError in the reference). */
int bocf_fit(bocf_ctx* ctx, const double* X, const double* Y, int N, int d, int m, int kernel_id,
             const double* variance, const double* lengthscale, const double* noise,
             int max_jitter_tries, double* jitter_out, double* lml_out);

/* Outputs with DIFFERENT kernel families: the reference's multi_outputGP takes a kernel list, one GPy kernel per output
 * (multi_outputGP.py:44-47 -> GPModel(kernel=...) gpmodel.py:50-61).  ids (m) are taken by the NEXT bocf_fit / bocf_infer / bocf_hmc
 * call with m outputs (hyper-sample-major like every per-output array) and override its kernel_id argument; the list is consumed by
 * that call (m = 0 clears a pending one).  The kernels that evaluate a covariance function are specialised per family at compile time:
 * the library issues one launch per run of equal ids. */
int bocf_set_kernel_ids(bocf_ctx* ctx, const int* ids, int m);

/* Same inputs, new targets Y (m,N): recomputes the mean-centring, alpha, the log-marginal and the cached
 * posterior mean at the training inputs (two GEMVs) -- what GP.set_XY(Y=...) costs the reference a full
 * inference for (GPy/core/gp.py:191-227). */
int bocf_update_targets(bocf_ctx* ctx, const double* Y, double* lml_out);

/* Rank-1 append of ONE observation x_new (d) with the complete new targets Y (m,N+1): borders the resident
 * factor U and its inverse R in O(N^2) instead of refitting in O(N^3) (the reference refits: cbo.py:363,419 ->
 * GP.set_XY).  Returns 0 on success; 1 when the caller must run bocf_fit instead (padding exhausted, i.e. N is
 * a multiple of 128; a jittered factor; or a non-positive new pivot, in which case the context is marked
 * unfitted). */
int bocf_append(bocf_ctx* ctx, const double* x_new, const double* Y, double* lml_out);

/* Gradients of the log marginal likelihood of the CURRENT fit w.r.t. the raw hyper-parameters: kernel variance
 * (m), lengthscales (m,d) (an isotropic kernel's single gradient is the sum over d), noise variance (m).  The
 * numerical core of hyper-parameter learning (GP.parameters_changed, GPy/core/gp.py:256-258): dL_dK and
 * dL_dthetaL of exact_gaussian_inference.py:61-63 + kern.update_gradients_full (stationary.py:191-214,
 * se.py:169-188).  Ky^-1 = R R^T replaces pdinv's dpotri.  Priors / transformations stay on the host. */
int bocf_lml_gradients(bocf_ctx* ctx, double* dvariance_out, double* dlengthscale_out, double* dnoise_out);

/* One hyper-parameter INFERENCE = bocf_fit's log-marginal + bocf_lml_gradients, in one call: what every step of
 * GPModel.updateModel's optimiser (gpmodel.py:115 -> paramz Model._objective_grads) and every leapfrog step of its HMC
 * (GPy/inference/mcmc/hmc.py:62-66) costs.  Arguments as bocf_fit; returns 0, the LAPACK-style info of a failed
 * factorization (> 0), or < 0.  Models with N <= 128 and d <= 16 run as ONE fused launch per jitter attempt (kernel
 * build, Cholesky, inverse, alpha, log-marginal, Ky^-1 and the gradient sums in one workgroup per output) and leave
 * no factor behind -- call bocf_fit before predicting; larger models run bocf_fit + bocf_lml_gradients.  Option
 * "fused_infer" = 0 forces the two-call path. */
int bocf_infer(bocf_ctx* ctx, const double* X, const double* Y, int N, int d, int m, int kernel_id, const double* variance,
               const double* lengthscale, const double* noise, int max_jitter_tries, double* jitter_out, double* lml_out,
               double* dvariance_out, double* dlengthscale_out, double* dnoise_out);

/* The HMC chain of GPModel.updateModel (gpmodel.py:117-118: HMC(model, stepsize).sample(num_samples, hmc_iters) ->
 * GPy/inference/mcmc/hmc.py:30-69 with M = I; 200 draws x 20 leapfrog steps = 4000 inferences per output and model update in the
 * reference) as ONE device launch, for the models the fused inference serves (N <= 128, d <= 16).  One workgroup per output runs its
 * whole chain: per leapfrog step the in-kernel inference (kernel matrix, jitchol ladder linalg.py:52-71, alpha, log-marginal,
 * hyper-gradients: exact_gaussian_inference.py:46-63, stationary.py:191-214), the objective -(log-marginal + log-prior) and its
 * gradient w.r.t. the optimizer array (paramz Model._objective_grads; Gamma priors priors.py:264-330, Logexp transform of paramz 0.9.1
 * as restated in bocf_amd/hyper.py), the momentum / position updates (hmc.py:62-66), and per draw the Hamiltonian and the
 * Metropolis test (hmc.py:45-59).  The HOST draws momenta and uniforms in the reference's RNG order and hands them in.
 *   theta (m, P) in/out, P = 2 + nls: [kern.variance, kern.lengthscale (nls = 1 isotropic | d ARD), Gaussian_noise.variance];
 *   fixed (m, P): 1 = constrain_fixed (not sampled);  prior Gamma(a, b) on every parameter;
 *   momenta (m, num_samples, P) / chains_out (m, num_samples, P): the free entries of a draw packed in front;  uniforms (m, num_samples);
 *   chains_out[i] = the state draw i started from, overwritten by the proposal when accepted (hmc.py:45-59);
 *   raise_on_failure = 1: a factorization that fails even with jitter (or parameters leaving the positive domain) stops that output's
 *   chain with status_out[j] = draw + 1 and the call returns 1 (hmc.py lets jitchol's LinAlgError propagate); 0: the proposal is
 *   rejected and the chain goes on.  A non-finite objective is rejected in both modes, as hmc.py:51-58 does.
 *   inferences_out: leapfrog-step inferences of the longest chain.  Leaves no factor behind (bocf_fit before predicting). */
int bocf_hmc(bocf_ctx* ctx, const double* X, const double* Y, int N, int d, int m, int kernel_id, double* theta, int nls, const int* fixed,
             double prior_a, double prior_b, const double* momenta, const double* uniforms, int num_samples, int hmc_iters, double stepsize,
             int max_jitter_tries, int raise_on_failure, double* chains_out, int* accepted_out, int* diverged_out, int* status_out,
             long long* inferences_out);

/* The same chain for models beyond the fused kernel (N > 128 or d > 16), STREAM-RESIDENT: GPy/inference/mcmc/hmc.py:30-69 as
 * GPModel.updateModel runs it (gpmodel.py:117-118).  Every leapfrog step (hmc.py:62-66) is the launch sequence of one inference
 * (bocf_fit's factorization + bocf_lml_gradients: exact_gaussian_inference.py:46-63, stationary.py:191-214) between two launches of a
 * small kernel that does what the host loop did per step -- momentum / position update, Logexp transform, Gamma priors (priors.py:264-330),
 * objective and gradient w.r.t. the optimizer array, and per draw the Hamiltonian and the Metropolis test (hmc.py:45-59); the host
 * enqueues and looks at one word every few draws.  Arguments as bocf_hmc.  jitchol's ladder (linalg.py:52-71) needs the host: when a
 * factorization meets a non-positive pivot (or parameters leave the positive domain) the chain stops with *draws_done_out = that draw
 * (< num_samples) and every output back at the draw's start; the caller runs that ONE draw with bocf_infer per step (ladder included) and
 * calls again for the remaining draws.  accepted_out / diverged_out / chains_out cover the completed draws.  Returns 0 or < 0.  Leaves no
 * usable factor behind (bocf_fit before predicting). */
int bocf_hmc_streamed(bocf_ctx* ctx, const double* X, const double* Y, int N, int d, int m, int kernel_id, double* theta, int nls,
                      const int* fixed, double prior_a, double prior_b, const double* momenta, const double* uniforms, int num_samples,
                      int hmc_iters, double stepsize, double* chains_out, int* accepted_out, int* diverged_out, int* draws_done_out,
                      long long* inferences_out);

/* Per-output status of the LAST bocf_fit / bocf_infer: info_out[j] = 0 when output j factorized (possibly on a jitter
 * rung), else the 1-based index of its first non-positive pivot on the last rung tried -- which outputs made jitchol give
 * up (GPy/util/linalg.py:56-71 raises for ONE matrix; here m are factorized together, so the caller needs to know which).
 * n must equal the m of that call. */
int bocf_last_fit_info(bocf_ctx* ctx, int* info_out, int n);

/* Test/inspection hooks: lower Cholesky factor L (N,N row-major) and alpha (N) of output j --
 * Posterior.woodbury_chol / woodbury_vector (posterior.py:132-170, 193-205). */
int bocf_get_factor(bocf_ctx* ctx, int j, double* L_out, double* alpha_out);
/* K(X,X) of output j as built on the device (N,N), without the diagonal noise: kern.K(X). */
int bocf_get_train_kernel(bocf_ctx* ctx, int j, double* K_out);

/* Test/inspection hook for the acquisition kernels alone: hand the context a posterior -- mean (m,C), var (m,C) exactly as
 * model.predict / posterior_variance would return it, and the posterior mean at N evaluated points mu_train (m,N) -- so
 * that bocf_acq_linear / bocf_set_mc_samples + bocf_acq_mc / bocf_select_topk run on it (the duck-typed Mock-model pattern
 * of the reference's own acquisition tests, GPyOpt/testing/acquisitions_tests/test_ei_acquisition.py:11-26).  Everything
 * that needs a factorization fails until the next bocf_fit. */
int bocf_set_posterior(bocf_ctx* ctx, int m, int C, int N, const double* mean, const double* var, const double* mu_train);

/* Upload the candidate batch X* (C,d); it stays resident in HBM until replaced. */
int bocf_set_candidates(bocf_ctx* ctx, const double* Xc, int C);

/* PREDICT on the resident candidates.  Replaces multi_outputGP.predict / posterior_mean /
 * posterior_variance[_noiseless] (multi_outputGP.py:138-200) -> PosteriorExact._raw_predict /
 * raw_posterior_mean / raw_posterior_variance (posterior.py:268-320):
 *   mean = K(X*,X) alpha + ymean;  var = sigma_f^2 - ||L^-1 K(X,X*)||^2 [+ noise] [clipped].
 * mean_out / var_out are (m,C) or NULL. */
int bocf_predict(bocf_ctx* ctx, int flags, double* mean_out, double* var_out);

/* The `full_cov=True` form of multi_outputGP.predict (multi_outputGP.py:138-149): every output's model returns its n x n predictive
 * covariance (PosteriorExact._raw_predict with full_cov, posterior.py:274-283: Kxx - tmp^T tmp; + noise on the diagonal,
 * gaussian.py:95-97; every ENTRY clipped at 1e-10, gpmodel_fixed_hyps.py:84-86 / gpmodel.py:145-147) and the wrapper keeps column 0
 * (cov[j,:] = tmp2[:,0], multi_outputGP.py:146-148).  cov0_out (m,C): cov0[j][i] = k_j(x_i, x_0) - k_j(x_i,X) Ky_j^-1 k_j(X, x_0)
 * [+ noise_j if i == 0 and BOCF_ADD_NOISE] [clipped at 1e-10 if BOCF_CLIP] for the resident candidates x_0 ... x_{C-1}; the n x n
 * matrix is never formed (one extra solve w = R (R^T k(X, x_0)) and a mean-shaped pass over the candidates). */
int bocf_predict_cov_column(bocf_ctx* ctx, int flags, double* cov0_out);

/* Input gradients of the posterior at the resident candidates, (m,C,d) each.  Replaces
 * multi_outputGP.posterior_mean_gradient / posterior_variance_gradient (multi_outputGP.py:284-306) ->
 * GP.posterior_mean_gradient / posterior_variance_gradient (GPy/core/gp.py:438-490) -> kern.gradients_X
 * (stationary.py:312-331, se.py:135-148); uses w = Ky^-1 k(X,x*) = R (R^T k*) instead of dpotri. */
int bocf_predict_gradients(bocf_ctx* ctx, double* dmean_out, double* dvar_out);

/* Posterior mean at the training inputs, (m,N): multi_outputGP.posterior_mean_at_evaluated_points
 * (multi_outputGP.py:176-180). */
int bocf_mean_at_train(bocf_ctx* ctx, double* out);

/* Closed-form acquisition of a LINEAR utility over the resident candidates.  Replaces
 * maEI._compute_acq/_marginal_acq/_marginal_best_so_far (maEI.py:38-54,81-98,129-136) and the
 * maPI / EI / PI twins.  theta (L,m), prob (L) or NULL (= plain mean over L sampled thetas,
 * maEI.py:52).  acq_out (C) or NULL (result stays on the device for bocf_select_topk). */
int bocf_acq_linear(bocf_ctx* ctx, int kind, const double* theta, const double* prob, int L, double* acq_out);

/* As bocf_acq_linear, plus d acq / dx (C,d): maEI._compute_acq_withGradients / _marginal_acq_with_gradient
 * (maEI.py:57-78,101-126; EI = (mu-best) Phi + sigma phi with scipy's norm.pdf/cdf) and the maPI twin
 * (maPI.py:56-76,96-121). */
int bocf_acq_linear_grad(bocf_ctx* ctx, int kind, const double* theta, const double* prob, int L, double* acq_out, double* dacq_out);

/* Upload the common-random-number normals W (S,m): uEI_noiseless.W_samples (uEI_noiseless.py:31). */
int bocf_set_mc_samples(bocf_ctx* ctx, const double* W, int S);

/* Monte-Carlo acquisition of a composite utility over the resident candidates.  Replaces
 * uEI_noiseless._compute_acq/_marginal_acq (uEI_noiseless.py:40-83) and uPI (uPI.py:43-86):
 *   acq(x) = sum_l p_l (1/S) sum_s hinge_or_indicator( U(theta_l, mu(x) + sigma(x) o W_s) - best_l ),
 *   best_l = max_i U(theta_l, mu(X_i)), sigma = sqrt(clipped posterior variance incl. noise).
 * theta (L,theta_dim) (may be NULL when theta_dim == 0). */
int bocf_acq_mc(bocf_ctx* ctx, int kind, int util_kind, const double* util_params, int n_util_params,
                const double* theta, int theta_dim, const double* prob, int L, double* acq_out);

/* Monte-Carlo EI with d acq / dx (C,d): uEI_noiseless._compute_acq_withGradients /
 * _marginal_acq_with_gradient (uEI_noiseless.py:118-170); dU/dy is the analytic derivative of the
 * device utility (the dfunc of the experiment scripts). */
int bocf_acq_mc_grad(bocf_ctx* ctx, int util_kind, const double* util_params, int n_util_params, const double* theta,
                     int theta_dim, const double* prob, int L, double* acq_out, double* dacq_out);

/* Selection on the last acquisition vector: indices of the k largest values, ties to the lowest
 * index -- np.argsort(-acq)[:k] of AnchorPointsGenerator.get (anchor_points_generator.py:59-61)
 * applied to AcquisitionBase.acquisition_function's -acq (GPyOpt/acquisitions/base.py:33-40).
 * idx_out (k) int64, val_out (k) or NULL. */
int bocf_select_topk(bocf_ctx* ctx, int k, long long* idx_out, double* val_out);

/* ---- multi-GPU: candidate shards, ONE collective.  One process per GPU, one context per process.  The
 * reference's own candidate parallelism is a pathos process pool over single candidates (uEI_noiseless.py:85-97); here rank
 * r scores the contiguous slice [lo_r, hi_r) of the batch against its resident fit and the ranks exchange only their k
 * local winners.  RCCL is bound at run time (dlopen librccl.so.1); errors come back as < 0 with the RCCL text in
 * bocf_last_error().
 *
 * bocf_comm_unique_id: rank 0 creates the 128-byte rendezvous id (ncclGetUniqueId); the host passes it to the other ranks by
 *   whatever channel launched them (torch.distributed broadcast, MPI, a file).
 * bocf_comm_init: collective over all ranks (ncclCommInitRank); the communicator is owned by the context.
 * bocf_comm_info: returns 1 if the context holds a communicator (and its world / rank), 0 if not. */
#define BOCF_COMM_ID_BYTES 128
int bocf_comm_unique_id(char* id_out);
int bocf_comm_init(bocf_ctx* ctx, const char* id_bytes, int world, int rank);
int bocf_comm_destroy(bocf_ctx* ctx);
int bocf_comm_info(bocf_ctx* ctx, int* world_out, int* rank_out);

/* Global selection over the sharded batch: local top-k of the last acquisition vector (device), packed with the global
 * indices lo + i into a 2 * world * k buffer of doubles (values | indices, -inf elsewhere: RCCL has no MAXLOC), ONE
 * ncclAllReduce(max) on the context's stream, merged on the device (value descending, index ascending) -- the
 * np.argsort(scores)[:num_anchor] of AnchorPointsGenerator.get (anchor_points_generator.py:59-61) on the WHOLE batch,
 * identical on every rank.  Without a communicator it is the single-rank selection with lo added.  Empty slots (fewer than
 * k candidates in total) come back as index -1, value -inf. */
int bocf_global_topk(bocf_ctx* ctx, int k, long long lo, long long* idx_out, double* val_out);

/* The same selection for a host that owns the collective (torch.distributed): bocf_topk_packed writes the packed buffer of
 * this rank into caller-provided DEVICE memory (2 * world * k doubles) and returns when it is complete; after the caller's
 * all-reduce(MAX) over that buffer bocf_merge_packed merges it on the device. */
int bocf_topk_packed(bocf_ctx* ctx, int k, long long lo, int world, int rank, void* device_buf);
int bocf_merge_packed(bocf_ctx* ctx, int k, int world, const void* device_buf, long long* idx_out, double* val_out);

/* Profiling of the dominant kernel (needs option "profile"=1): accumulated HIP-event time in
 * ms and launch count since the last reset; algorithmic flops of those launches. */
int bocf_profile_read(bocf_ctx* ctx, double* ms_out, long long* launches_out, double* flops_out, int reset);

/* Named phases (option "profile" = 1): accumulated HIP-event time on the context's stream and number of brackets since the
 * last reset.  Names: "kbuild" (K(X,X) build), "cholesky", "inverse" (mirror + triangular inverse), "alpha" (alpha,
 * log-marginal, train mean) of bocf_fit; "cross" (K(X,X*) + mean), "acq", "topk" of the acquisition call. */
int bocf_profile_phase(bocf_ctx* ctx, const char* name, double* ms_out, long long* count_out, int reset);

/* Counters and facts about the context (diagnostics; none of them changes a result).  Names: "sched_timeouts" = how often a
 * multi-stream factorization schedule ran into its 0.2 s dependency time-out and the attempt was redone on the single-stream
 * schedule (then "gated_schedules_off" = 1 for the rest of the context's life); "last_schedule" = schedule of the last
 * factorization (0 single stream, 2 reserved CUs);
 * "early_inverse";
 * "cu_masks_ok"; "comm_world" = ranks of the context's RCCL communicator (0 = none); "kstar_workspace_bytes". */
int bocf_get_stat(bocf_ctx* ctx, const char* name, long long* value_out);

/* The option table of bocf_set_option, readable without a GPU: bocf_option_count() entries; bocf_option_info gives name, accepted
 * range, kind (0 = speed only: same result up to rounding; 1 = documented semantics; 2 = probe / test hook -- present ONLY in the
 * -DBOCF_PROBES build libbocf_hip_probes.so that tools/ and some tests load, never in libbocf_hip.so) and a one-line description;
 * bocf_option_check(name, value) = 0 if bocf_set_option would accept the pair, else < 0 (text in bocf_last_error()). */
int bocf_option_count(void);
int bocf_option_info(int index, const char** name_out, long long* lo_out, long long* hi_out, int* kind_out, const char** what_out);
int bocf_option_check(const char* name, long long value);

/* Block until the context's stream is idle. */
int bocf_sync(bocf_ctx* ctx);

/* The acquisition optimiser's inner loop (host arithmetic, no context): replaces the 16 sequential scipy.optimize.fmin_l_bfgs_b runs of
 * GPyOpt/optimization/optimizer.py:283-354 (OptLbfgs: maxiter 500, factr 1e6; OptLbfgs2: maxiter 50, factr 1e5, pgtol 1e-15) behind
 * acquisition_optimizer.py:66-78 (one run per anchor point) by ONE batched run: all A starts advance together, f_df is called once per
 * trial step on the rows still running -- with the device acquisitions that is one device pass per step for all anchors.  L-BFGS-B's
 * stopping tests (max |projected gradient| <= pgtol, (f_k - f_k+1) / max(|f_k|, |f_k+1|, 1) <= factr * eps, maxiter iterations, maxfun >= 0
 * trial points per row), its first move and its curvature test; m curvature pairs per row, Armijo constant c1 and at most max_ls trial
 * steps per iteration along the projection arc.  f_df(user, Z (n x d), rows (n: which start each row of Z belongs to), n, d, f_out (n),
 * g_out (n x d)) returns 0, or non-zero to abort (the call then returns 2).  X0, X_out: A x d; lo, hi: d (infinite allowed);
 * F_out: A; calls_out[2] = callbacks, points evaluated; iters_out[A] = iterations per start.  Returns 0, 1 (bad argument) or 2. */
typedef int (*bocf_fdf_callback)(void* user, const double* Z, const int* rows, int n, int d, double* f_out, double* g_out);
int bocf_lbfgsb_batched(bocf_fdf_callback f_df, void* user, const double* X0, int A, int d, const double* lo, const double* hi, int maxiter,
                        int m, double factr, double pgtol, int max_ls, double c1, int maxfun, double* X_out, double* F_out,
                        long long* calls_out, int* iters_out);

#ifdef __cplusplus
}
#endif
#endif
