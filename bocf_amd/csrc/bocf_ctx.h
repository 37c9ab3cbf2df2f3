// Context of one GPU (opaque bocf_ctx of include/bocf_hip.h) and the error plumbing shared by capi.hip and comm.hip.
#pragma once
#include "bocf_internal.h"
#include "../../include/bocf_hip.h"

#include <map>
#include <string>
#include <utility>
#include <vector>

int bocf_fail(const char* what, const char* detail);      // records bocf_last_error(), returns -1
void bocf_set_error(const char* text);                    // records bocf_last_error() verbatim (positive LAPACK-style returns)
int bocf_launch_status();
int bocf_run_cholesky(bocf_ctx* c);                       // capi_chol.hip: blocked Cholesky of all outputs, schedule by size / option
int bocf_run_trtri(bocf_ctx* c, bool early_done);         // capi_chol.hip: R = U^-1 (the part the factorization did not already start)
int bocf_comm_broadcast(bocf_ctx* c, double* buf, size_t count, int root);   // comm.hip: ncclBroadcast on the context's stream
int bocf_comm_group(bool start);                                             // ncclGroupStart / ncclGroupEnd
int bocf_comm_abort(bocf_ctx* c);                                            // ncclCommAbort: peers fail instead of blocking
int bocf_comm_allreduce_sum(bocf_ctx* c, double* buf, size_t count);        // in place, on the context's stream
#define fail bocf_fail
#define HIPCHK(expr)                                                         \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) return fail(#expr, hipGetErrorString(e_));         \
  } while (0)
#define LAUNCHCHK()                       \
  do {                                    \
    if (bocf_launch_status()) return -1;  \
  } while (0)

// host array of the per-output kernel ids of the resident model, or nullptr when every output uses c->kernel_id
#define BOCF_KIDS(c) ((int)(c)->kernel_ids.size() == (c)->m && (c)->m > 0 ? (c)->kernel_ids.data() : nullptr)
static inline int round_up(int x, int q) { return (x + q - 1) / q * q; }
static inline int nsplit_for(int Np, int Cpad, int m) {
  const int blocks = ((Cpad + 511) / 512) * m;        // cross_kernel: 256 threads x 2 columns per workgroup
  int ns = 2048 / (blocks > 0 ? blocks : 1);
  if (ns < 1) ns = 1;
  const int maxs = Np / BOCF_TILE;
  if (ns > maxs) ns = maxs;
  return ns;
}



struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail("hipMalloc", hipGetErrorString(e));
    cap = bytes;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct bocf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // cross-kernel stream (overlaps the VALU/HBM-bound K* build with the MFMA-bound GEMM)
  // small batches (the single points and 16-point groups of the acquisition optimiser): candidates go up and results come back through pinned
  // staging buffers -- asynchronous copies, ONE stream synchronisation per call instead of one per pageable copy
  void* pin_in = nullptr; void* pin_out = nullptr;
  size_t pin_in_cap = 0, pin_out_cap = 0;
  hipEvent_t ev_pin = nullptr;     // the upload out of pin_in has completed
  void* fit_pin = nullptr; size_t fit_pin_cap = 0;     // status words + log-marginal of a fit
  void* up_pin = nullptr; size_t up_pin_cap = 0, arena_used = 0;   // pinned arena of a fit's host-to-device copies (X, hyper-parameters, targets, jitter)
  hipEvent_t ev_start = nullptr;
  std::vector<hipEvent_t> ev_parts;
  std::vector<hipEvent_t> ev_chol;  // lookahead Cholesky: events per panel
  // Reserved-CU lookahead (run_cholesky): the serial chain of diagonal-block factorizations runs on a stream whose CU mask
  // holds `res_cus` compute units that NO other stream of the factorization may use (the trailing updates run on streams
  // masked to the complement), so a diagonal block never waits for a CU to drain and never shares one.
  hipStream_t s_res = nullptr, s_hi = nullptr, s_bulk = nullptr;
  DevBuf chol_flags;         // device-side dependency counters of the reserved-CU schedule (+ the timeout word)
  int chol_flags_used = 0;
  int chol_err_off = 0;      // index of the time-out word inside chol_flags (set by the schedule that used them)
  int team_fit = -1;         // one-launch factorization + inverse by resident workgroup teams (chol_team.hip): -1 = by size (2..24 panels), 0 / 1 = never / whenever it applies
  int team_panels = 6;       // panels per team launch where teams work in groups (the first block rows of the hybrid schedule; team_fit = 1 without hybrid), each followed by ONE trailing update with K = 128 x that
  int flags_device_zeroed = 0;   // the caller's kernels zero the team schedule's counters in front of every factorization (stream-resident HMC)
  int want_kinv = 0;         // the caller is an INFERENCE (bocf_lml_gradients follows): a schedule that can, leaves Ky^-1 in the T scratch
  int kinv_done = 0;         // ... and did
  int team_hybrid = 2;       // more than team_whole_max panels: the first block rows by team launches of team_panels panels + trailing updates (1: by the launched schedule), ONE team launch (Cholesky + inverse) for the rest; 0: team_fit = 1 means panel groups throughout
  int team_tail_share = 5;   // hybrid schedule: eighths of the compute units the tail's teams take (the rest is for the early inverse underneath)
  int team_whole_max = 24;   // panels up to which ONE team launch factors and inverts everything (beyond: the hybrid schedule)
  int team_crit_load = 4;    // teams: the workgroups that stream the critical units carry nothing else while the others get by with <= this many units each
  int team_stream = 1;       // teams: U[p][p+1] and the last row of A[p+1][p+1] are formed 16 rows at a time underneath potrf(p) by a workgroup of their own
  int inverse_done = 0;      // the factorization schedule already produced R and R^T (team schedule)
  int ncu = 0;               // compute units of the device (read once)
  unsigned long long* team_tl = nullptr;   // probes build: task timeline of the team kernel (tools/team_timeline.py)
  int res_cus = 0;           // CUs currently reserved by s_res (0 = streams not created)
  int cu_masks_ok = 1;       // cleared when hipExtStreamCreateWithCUMask is refused: the single-stream schedules are used
  int lookahead = -1;        // -1: by size; 0: single stream; 1: two-stream lookahead of round 1 (only with aggregate = 1); 2: reserved-CU schedule
  // inverse overlapped with the factorization: the part that needs only the first h block rows runs on s_inv
  int overlap_inverse = -1;  // -1 = by size (from N = 4096 with at least two outputs: -4 % at 4096, -6 % at 6144, -2.5 % at 8192; neutral below), 0 / 1 = off / on
  hipStream_t s_inv = nullptr;
  hipEvent_t ev_half = nullptr, ev_inv_early = nullptr;
  int early_inverse_started = 0;
  void* zeroed_R = nullptr; void* zeroed_RT = nullptr; int zeroed_Np = 0, zeroed_m = 0;
  int gated_off = 0;         // latched by bocf_fit when a device-side dependency timed out: single-stream schedules from then on
  long long sched_timeouts = 0;   // how often that happened (bocf_get_stat "sched_timeouts")
  int sched_retry = 0;       // the next factorization attempt is the redo of one that timed out: single-stream
  long long fits_done = 0;   // successful bocf_fit calls of this context
  int last_schedule = 0;     // schedule of the last factorization: 0 single stream, 2 reserved CUs
  int sched_m = 0;           // > 0: choose the schedule as for this many outputs (the helper context of an output-sharded fit)
  int force_sched_timeout = 0, force_cu_count = 0;   // test hooks (BOCF_PROBES builds only)
  int lookahead_min_nb = 8;  // reserved-CU lookahead from this many 128-panels on
  int aggregate = 0;         // panels per trailing update of the blocked Cholesky (0 = by size, 1 = classic right-looking)
  int data_N = 0, data_d = 0, data_m = 0;   // shape of the X / Y resident on the device
  int fused_infer = 1;       // bocf_infer: one fused launch for N <= 128, d <= 16
  int reuse_data = 0;        // next bocf_fit calls: X, Y (and N, d, m) are those of the previous fit -- only the hyper-parameters change
  int skip_mu_train = 0;     // do not refresh the posterior mean at the training inputs (HMC / optimiser inferences never read it)
  DevBuf gpart, gout;        // bocf_lml_gradients scratch
  DevBuf hmc_buf;            // bocf_hmc: parameters, momenta, uniforms, chains, counters
  double* infer_out = nullptr;    // host-mapped result block of the fused inference (the kernel writes it over PCIe: no D2H copy)
  size_t infer_out_cap = 0;
  int overlap = 0;                 // measured: no gain (the K* build slows the co-running GEMM by as much as it hides)
  // ---- fit state
  bool fitted = false;
  bool canned = false;       // bocf_set_posterior: mean / var / train mean were given by the host (acquisition kernels only)
  int N = 0, Np = 0, d = 0, m = 0, kernel_id = 0;
  long xs_stride = 0;        // per-output stride of Xs (capacity Np rows so that observations can be appended)
  std::vector<int> kernel_ids;   // per-output kernel family of the resident model when the outputs differ (else empty: all kernel_id)
  std::vector<int> pending_ids;  // bocf_set_kernel_ids: taken by the next bocf_fit / bocf_infer / bocf_hmc with as many outputs
  std::vector<KernHyp> hyp;
  std::vector<double> jitter;
  std::vector<int> last_info;  // per-output LAPACK-style info of the last bocf_fit / bocf_infer attempt (0 = factorized)
  DevBuf R32;                // fp32 copy of R for the fp32 variance contraction (option predict_f32)
  bool r32_valid = false;
  int predict_f32 = 0;
  // int8 (Ozaki) variance contraction (option predict_i8): digit fragments of R and their column exponents (per fit), of K* (per chunk), the
  // kernels' variance exponents
  DevBuf Ri8, Ri8e, Ki8, Ki8e;
  bool ri8_valid = false;
  int predict_i8 = 0, i8_group = 0;    // i8_group: 0 = XCD blocks of 4 row-tile pairs x 8 column tiles, g >= 1 = bands of g pairs (speed only)
  DevBuf X, Xs, S, R, RT, E, ET, T, yc, tvec, rvec, dvec, alpha, lml, jit, hypd, info, mu_train;
  // ---- candidates
  int C = 0;
  DevBuf Xc;
  // ---- workspace
  long chunk = 65536;
  long workspace_mb = 24576; // cap of the per-pass K* / V workspace
  DevBuf Kstar, meanpart, sumsq, mean, var, acq, Vbuf, dmean, dvar, dacq, Vs, Ws;
  int pred_cap = 0;          // columns allocated in mean/var/acq
  // ---- acquisition parameters
  DevBuf theta, prob, best, params, Wt;
  std::vector<double> last_params;   // host copy of what theta/prob/params hold (skip identical re-uploads)
  long long mu_epoch = 0;            // bumped whenever mu_train changes (fit, target refresh, append, canned posterior)
  long long best_epoch = -1; int best_sig = -1;   // what c->best was computed from: mu_epoch and (linear, utility kind, group); reset when the parameters are re-uploaded
  int S_mc = 0;
  bool have_acq = false;
  DevBuf blk_idx, blk_val, out_idx, out_val;
  // ---- profiling of the dominant kernel
  bool profile = false;
  double test_diag_shift = 0.0;
  int prefetch1 = 0;
  int merge_x3 = 1;          // the second product of an inverse merge in the three-buffer triangular kernel: 0 never, 1 from 4096 rows, 2 whenever possible
  int potrf_scalar = 0;      // probes build: 11..14 = timing-only variants of the diagonal-block kernel
  int trsm_wave = 1;         // row solves of the factorization through the wave-level single-tile kernel (0: the 128 x 128 GEMM kernel)
  int kstar_valu_probe = 0;  // timing-only experiment (gemm_f64.hip, VPROBE)
  int small_path = 1;        // GEMV-shaped path for <= 16 candidates
  int hyper_samples = 1;     // H: the m outputs are H groups (hyper-samples, group-major) of m / H model outputs
  int acq_hyper_samples = 0; // hyper-samples the acquisitions average over (0 = all; the reference uses min(10, H), maEI.py:35)
  int best_group = -1;       // -1: each hyper-sample's own best-so-far (maEI.py:88); >= 0: that group's for every h (uEI_noiseless.py:66)
  int swizzle = -1;          // variance GEMM tiling: -1 = by size (256-row three-buffer kernel from 2048 candidates per pass), 0 = 128-row tiles,
                             // 258 = 256-row tiles; probes build also 1 / 100+RT (tile orders measured slower), 256 / 257 (two-buffer kernel)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  double prof_flops = 0.0;
  std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> phases;   // named phases (bocf_profile_phase)
  // ---- multi-GPU (comm.hip): RCCL communicator of this rank, buffers of the one collective of the path
  int shard_fit = 0;         // option: bocf_fit factorizes only this rank's share of the outputs and exchanges the inverse factors
  int shard_fit_simulate = 0;   // test hook: G > 0 = one process plays all G ranks in turn (no collectives)
  bool sharded = false;      // the current fit holds R / R^T / alpha only (no upper factor)
  bocf_ctx* shard_helper = nullptr;
  DevBuf shard_meta;
  void* comm = nullptr;      // ncclComm_t
  int world = 1, rank = 0;
  DevBuf pack, gidx, gval;
};


// HIP-event bracket of a named phase on the context's stream (only with option "profile" = 1; otherwise free)
struct PhaseTimer {
  bocf_ctx* c;
  const char* name;
  hipEvent_t e0 = nullptr;
  PhaseTimer(bocf_ctx* ctx, const char* n) : c(ctx), name(n) {
    if (!c->profile) return;
    if (hipEventCreate(&e0) != hipSuccess) { e0 = nullptr; return; }
    (void)hipEventRecord(e0, c->stream);
  }
  void stop() {
    if (!e0) return;
    hipEvent_t e1 = nullptr;
    if (hipEventCreate(&e1) == hipSuccess) {
      (void)hipEventRecord(e1, c->stream);
      c->phases[name].emplace_back(e0, e1);
    } else {
      (void)hipEventDestroy(e0);
    }
    e0 = nullptr;
  }
  ~PhaseTimer() { stop(); }
};
