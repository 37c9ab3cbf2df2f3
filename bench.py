"""Benchmark of the hot path: one STEP = one batch acquisition call over the candidate batch = predict (cross kernel
K(X,X*), N^2 C variance contraction, mean) + Monte-Carlo uEI + top-16 selection [+ ONE all-reduce over ranks].
Default workload = BASELINE.json configs[2]: m=4 RBF-ARD, N=4096, d=8, S=1024 MC samples, C=65536 candidates, fp64.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0:
  value / ms_per_step         K steps with the inputs (X*, W) resident in HBM, total wall time between two fences, max over ranks
  ms_per_step_median          median of the K per-step times of that region
  transfers_included          the SAME K steps once more with the H2D of X* and the D2H of the scores inside every step
                              (SURVEY 8d Metric 1 as worded: "host<->device transfer of X* and acq included; median")
  roofline                    dominant kernel (variance GEMM): algorithmic flop / HIP-event time, in-run
  roofline_fit                K(X,X) build GB/s vs HBM peak and Cholesky + inverse TFLOP/s vs fp64-MFMA peak, HIP events in-run
  cpu_baseline                the oracle (NumPy/SciPy port) on a bounded sample of the same workload, plus the reference's own
                              loop structure (uEI_noiseless.py:63-83) timed on a small sample and its survey-time figure
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_MFMA_PEAK_TOPS = 5033.0   # v_mfma_i32_16x16x64_i8: 256 CU x 4 SIMD x 32768 op / 16 cycles x 2.4 GHz (a register-only loop sustains 4.9 POP/s)
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md, Peak FP32 matrix)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X vendor peak FP64 matrix; v_mfma_f64_16x16x4_f64
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def pmc_traffic(N, m, c_local, f32=False):
    """HBM bytes per launch of the variance GEMM from the committed rocprofv3 PMC passes of THIS command
    (profiles/*/gemm_traffic*.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs; KiB units;
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  (bytes, file) or (None, None)."""
    best = (None, None)
    pdir = os.path.join(ROOT, "profiles")
    for r in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not os.path.isdir(os.path.join(pdir, r)):
            continue
        for name in sorted(os.listdir(os.path.join(pdir, r))):
            if not (name.startswith("gemm_traffic") and name.endswith(".json")):
                continue
            t = json.load(open(os.path.join(pdir, r, name)))
            if (t.get("N"), t.get("m"), t.get("C_local"), bool(t.get("f32", False))) == (N, m, c_local, bool(f32)):   # later directories win
                best = ((2.0 * t["FETCH_SIZE_KiB"] + t["WRITE_SIZE_KiB"]) * 1024.0, "profiles/%s/%s" % (r, name), t.get("kernel"), t.get("git_head"),
                        t.get("source_sha"))
    return best if len(best) == 5 else (None, None, None, None, None)


def kernel_source_sha(f32=False):
    """git blob hash of the translation unit that holds the variance contraction: a committed counter file is valid only for the
    source it was taken on (a re-tiled kernel of the same NAME would otherwise be priced with old bytes)."""
    import hashlib
    path = os.path.join(ROOT, "bocf_amd", "csrc", "gemm_f32.hip" if f32 else "gemm_f64.hip")
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--S", type=int, default=1024)
    ap.add_argument("--C", type=int, default=65536)
    ap.add_argument("--kernel", default="rbf")
    ap.add_argument("--seed", type=int, default=1237)
    ap.add_argument("--noise", type=float, default=1e-6)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 5],
                    help="BASELINE.json config preset (1-based as in SURVEY 8d): 2 = N1024 d6 S256 C8192; 3 = headline; "
                         "5 = m8 Matern52 N8192 d12 S4096 noise 1e-4 (add --f32 for the fp32 contraction it names)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="candidates in the CPU-baseline sample")
    ap.add_argument("--f32", action="store_true", help="fp32 variance contraction (option predict_f32; BASELINE configs[4] arithmetic)")
    ap.add_argument("--i8", action="store_true", help="variance contraction in exact int8 digit products (option predict_i8: six radix-254 digits per operand column, fp64 recombination)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="bocf_set_option passthrough (A/B experiments)")
    ap.add_argument("--comm", default="auto", choices=["auto", "native", "torch"],
                    help="carrier of the one collective at N>1: the context's own RCCL communicator (native), torch.distributed's "
                         "all-reduce over a device-packed buffer (torch), or native with torch as the fallback (auto)")
    ap.add_argument("--shard-fit", action="store_true", help="N>1: every rank factorizes only its contiguous share of the outputs, inverse factors broadcast over RCCL (option shard_fit)")
    ap.add_argument("--check", action="store_true", help="parity-check a slice against the oracle before timing")
    return ap.parse_args()


PRESETS = {2: dict(N=1024, d=6, m=4, S=256, C=8192, kernel="rbf", seed=1236),
           5: dict(N=8192, d=12, m=8, S=4096, C=65536, kernel="matern52", seed=1239, noise=1e-4)}


def kbuild_bytes(Np, m):
    """Bytes the K(X,X) build writes: 64 x 512 tiles on/above the diagonal only (build_train_kernel), 8 B per element."""
    total = 0
    for rb in range(Np // 64):
        for cb in range((Np + 511) // 512):
            if cb * 512 + 511 >= rb * 64:
                total += 64 * min(512, Np - cb * 512)
    return 8.0 * m * total


def main():
    a = parse()
    for k, v in PRESETS.get(a.config, {}).items():
        setattr(a, k, v)
    if a.gpus > 1 and "RANK" not in os.environ:
        # launched without torchrun: start one rank per GPU as CHILD processes (nothing has touched the GPU yet) and
        # exit with their code -- the driver itself launches N>1 through torch.distributed.run directly
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29555"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch                                      # before libbocf_hip.so: one HIP runtime in the process
    dist = None
    if world > 1 or os.environ.get("BOCF_FORCE_DIST"):   # BOCF_FORCE_DIST: exercise the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    import bocf_amd as B
    from bocf_amd.distributed import device_global_topk, init_native_comm, shard_bounds
    from bocf_amd.synthetic import synthetic_problem

    p = synthetic_problem(a.N, a.d, a.m, a.C, a.S, a.seed, noise=a.noise)
    kcls = {"rbf": B.kern.RBF, "se": B.kern.SE, "matern52": B.kern.Matern52}[a.kernel]
    kern = [kcls(a.d, variance=p["variances"][j], lengthscale=p["lengthscales"][j], ARD=True) for j in range(a.m)]
    model = B.multi_outputGP(a.m, kernel=kern, noise_var=p["noise"], fixed_hyps=True, device=local_rank)
    lib = B._ffi.load()
    handle = model._context().handle

    if a.f32:
        model.set_option("predict_f32", 1)
    if a.i8:
        model.set_option("predict_i8", 1)
    for kv in a.option:
        model.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    carrier = "none"
    if dist is not None:
        carrier = "torch"
        if a.comm in ("auto", "native"):
            try:
                init_native_comm(model)
                carrier = "native"
            except Exception as e:                     # same packing, same merge, same result through torch's all-reduce
                if a.comm == "native":
                    raise
                sys.stderr.write("rank %d: native RCCL communicator unavailable (%s); using torch.distributed's all-reduce\n" % (rank, e))
        flag = torch.tensor([1.0 if carrier == "native" else 0.0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)    # every rank must use the same carrier
        if flag.item() < 1.0 and carrier == "native":
            lib.bocf_comm_destroy(handle)
            carrier = "torch"
    if a.shard_fit:
        model.set_option("shard_fit", 1)

    def phase(name, reset=1):
        ms, n = ctypes.c_double(), ctypes.c_longlong()
        B._ffi.check(lib.bocf_profile_phase(handle, name.encode(), ctypes.byref(ms), ctypes.byref(n), reset), "bocf_profile_phase")
        return ms.value, n.value

    # ---- GP fit (metric 2): K build + Cholesky + inverse factor + alpha for all m outputs, incl. H2D
    model.incremental = False                    # time the FULL fit (an unchanged X would otherwise only refresh alpha)
    for _ in range(4):                           # warm-up: allocations, stream creation, and the clock ramp of a process that has just started
        model.updateModel(p["X"], p["Y"])        # (successive fits of a fresh process read 7.05, 6.67, 6.59, 6.50, 6.52 ms)
    fit_ms = []
    for _ in range(7):                           # (median of seven, as the phase timers below)
        t0 = time.perf_counter()
        model.updateModel(p["X"], p["Y"])
        fit_ms.append((time.perf_counter() - t0) * 1e3)
    fit_ms = float(np.median(fit_ms))
    model.set_option("profile", 1)               # more fits with HIP events around the phases: one untimed (it creates the events), then three
    model.updateModel(p["X"], p["Y"])
    for name in ("kbuild", "cholesky", "inverse", "alpha"):
        phase(name)
    # seven fits read one by one, the MEDIAN reported: about one fit in twelve of a long-running process is 0.5-0.8 ms slower (the host stalls
    # once in the middle of enqueuing it and the GPU runs dry: seen as 4.57 / 4.57 / 4.86 / 4.58 ms for successive groups of three)
    nfit = 7
    per_fit = []
    for _ in range(nfit):
        model.updateModel(p["X"], p["Y"])
        per_fit.append({name: phase(name) for name in ("kbuild", "cholesky", "inverse", "alpha")})
    model.set_option("profile", 0)
    ph = {name: (float(np.median([f[name][0] for f in per_fit])) * nfit, sum(f[name][1] for f in per_fit)) for name in ("kbuild", "cholesky", "inverse", "alpha")}
    ci_per_fit = [f["cholesky"][0] + f["inverse"][0] for f in per_fit]
    Np = (a.N + 127) // 128 * 128
    kb_ms = float(np.median([f["kbuild"][0] / max(1, f["kbuild"][1]) for f in per_fit]))
    ci_ms = float(np.median(ci_per_fit))
    m_local = a.m if not a.shard_fit else len(range(rank, a.m, world))
    kb_bytes = kbuild_bytes(Np, m_local)
    ci_flops = 2.0 * m_local * float(a.N) ** 3 / 3.0          # N^3/3 (Cholesky) + N^3/3 (triangular inverse) per output
    roofline_fit = {
        "kbuild": {"kernel": "build_train_kernel (K(X,X), tiles on/above the diagonal)", "bound": "hbm", "bytes_per_launch": kb_bytes,
                   "launch_ms": kb_ms, "achieved": kb_bytes / (kb_ms * 1e-3) / 1e9 if kb_ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": (kb_bytes / (kb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kb_ms > 0 else 0.0,
                   "jitter_attempts_per_fit": ph["kbuild"][1] / nfit},
        "cholesky_inverse": {"kernels": "blocked Cholesky (diagonal-block kernel + fp64-MFMA panel solves / trailing updates) + triangular inverse",
                             "bound": "mfma", "algorithmic_flops": ci_flops, "ms": ci_ms, "ms_per_fit": ci_per_fit, "ms_definition": "median of %d fits" % nfit,
                             "cholesky_ms": ph["cholesky"][0] / nfit,
                             "inverse_ms": ph["inverse"][0] / nfit, "alpha_lml_trainmean_ms": ph["alpha"][0] / nfit,
                             "achieved": ci_flops / (ci_ms * 1e-3) / 1e12 if ci_ms > 0 else 0.0, "peak": FP64_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": (ci_flops / (ci_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS) if ci_ms > 0 else 0.0},
    }

    theta = np.array([[0.2 * (j + 1) for j in range(a.m)]])
    U = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), device="neg_sq_dist")
    acq = B.uEI_noiseless(model, None, utility=U)
    acq.W_samples = p["W"]

    lo, hi = shard_bounds(a.C, world, rank)
    Xloc = np.ascontiguousarray(p["Xc"][lo:hi])
    # inputs resident in HBM before the timed region: candidates and the MC normals
    model._set_candidates(Xloc)
    model.set_mc_samples(acq.W_samples)
    kind = U.device_kind()
    one = np.ones(1)

    def select():
        if dist is None:
            return model.select_topk(16)
        return device_global_topk(model, lo, 16)     # local top-16 -> packed on the device -> ONE all-reduce(MAX) -> merged on the device

    def step():                                      # inputs resident
        model._acq_mc_resident(B._ffi.ACQ_EI, kind, None, theta, one, None, fetch=False)
        return select()

    def step_with_transfers():                       # + H2D of this rank's X* slice, + D2H of its scores
        n = model._set_candidates(Xloc)
        scores = model._acq_mc_resident(B._ffi.ACQ_EI, kind, None, theta, one, n, fetch=True)
        return select(), scores

    if a.check and rank == 0:
        from oracle import cpu_ref as R      # checker only (--check)
        ref = R.MultiOutputGPRef(a.kernel, p["variances"], p["lengthscales"], p["noise"])
        ref.updateModel(p["X"], p["Y"])
        n = min(256, hi - lo)
        got = acq._compute_acq(Xloc[:n])
        want, _, _ = R.batch_uEI(ref, Xloc[:n], p["W"], "neg_sq_dist", theta, np.ones(1), "EI")
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-10)
        model._set_candidates(Xloc)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        per = []
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            s0 = time.perf_counter()
            out = fn()
            per.append(time.perf_counter() - s0)
        fence()
        dt = time.perf_counter() - t0
        ranks = None
        if dist is not None:
            # the whole region's time of every rank (max = the reported one) and every rank's median step: one gather, so that a slow rank,
            # a slow link or an unbalanced shard is visible in the first multi-GPU run
            mine = torch.tensor([dt, float(np.median(per)), float(np.min(per)), float(np.max(per))], device="cuda", dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(allr, mine)
            ranks = [[float(x) for x in r.tolist()] for r in allr]
            dt = max(r[0] for r in ranks)
        return dt, per, out, ranks

    for _ in range(a.warmup):
        step()
    model.set_option("profile", 1)
    lib.bocf_profile_read(handle, None, None, None, 1)
    for name in ("cross", "acq", "topk", "allreduce"):
        phase(name)
    dt, per, (top_idx, top_val), ranks = timed(step)
    ms, launches, flops = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
    lib.bocf_profile_read(handle, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(flops), 1)
    ph2 = {name: phase(name) for name in ("cross", "acq", "topk", "allreduce")}
    model.set_option("profile", 0)
    step_with_transfers()
    dt2, per2, _, _ = timed(step_with_transfers)
    # the same steps once more with the variance contraction in exact int8 digit products (option predict_i8): reported NEXT TO the fp64 line,
    # never as `value`
    i8_block = None
    if not a.f32 and not a.i8 and dist is None:
        model.set_option("predict_i8", 1)
        step()
        dt3, per3, (top_idx8, top_val8), _ = timed(step)
        model.set_option("predict_i8", 0)
        i8_block = {"what": "the same K steps with option predict_i8 = 1: V = L^-1 K* from six radix-254 int8 digits per operand column (47.9 bits), 21 exact int8 products "
                            "(v_mfma_i32_16x16x64_i8), fp64 recombination and sum of squares; fit, mean, acquisition and selection unchanged (fp64)",
                    "value": float(a.C) * a.S * a.steps / dt3, "ms_per_step": dt3 / a.steps * 1e3, "ms_per_step_median": float(np.median(per3)) * 1e3,
                    "speedup_over_fp64_step": dt / dt3,
                    "top16_identical_to_fp64": bool(np.array_equal(np.asarray(top_idx8), np.asarray(top_idx))),
                    "max_abs_diff_of_top16_acquisition_values": float(np.abs(np.asarray(top_val8) - np.asarray(top_val)).max())}

    if rank == 0:
        evals = float(a.C) * a.S * a.steps
        gemm_ms = ms.value / max(1, launches.value)
        gemm_flops = flops.value / max(1, launches.value)          # algorithmic: m N^2 C_local per launch
        ach = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        peak = FP32_MFMA_PEAK_TFLOPS if a.f32 else FP64_MFMA_PEAK_TFLOPS
        if a.i8:                                       # priced in int8 operations: 21 digit products per fp64 multiply-add pair
            gemm_flops *= 21.0
            ach = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
            peak = INT8_MFMA_PEAK_TOPS
        traffic, traffic_src, traffic_kernel, traffic_head, traffic_sha = pmc_traffic(a.N, a.m, hi - lo, a.f32)
        # the 256-row three-buffer kernels take batches from 2048 candidates per pass when the padded N is a multiple of 256 (capi.hip, gemm_f32.hip)
        big_tiles = (hi - lo) >= 2048 and ((a.N + 127) // 128 * 128) % 256 == 0 and not any(o.startswith("swizzle=") for o in a.option)
        kernel_name = (("gemm_tn_f32_sumsq256x3_kernel" if big_tiles else "gemm_tn_f32_sumsq_kernel") if a.f32
                       else ("gemm_tn_f64_sumsq256x3_kernel" if big_tiles else "gemm_tn_f64_kernel<1>"))
        if a.i8:
            kernel_name = "slice_operand_kernel<6> + var_i8_kernel<6>"
            traffic_src = None
        # a committed counter file is only valid for the kernel SOURCE it was taken on: tools/summarize_profiles.py stamps the git blob hash
        # of gemm_f64.hip / gemm_f32.hip; a file without the stamp, or with another hash, is stale (traffic = null)
        traffic_stale = bool(traffic_src and ((traffic_kernel and kernel_name not in traffic_kernel) or traffic_sha != kernel_source_sha(a.f32)))
        rccl_ranks = model._context().stat("comm_world")
        out = {
            "metric": "acquisition evals/sec (candidates x MC-samples/sec), uEI_noiseless batch call; GP-fit ms alongside",
            "value": evals / dt, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "ms_per_step_median": float(np.median(per)) * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("f32 (variance contraction) / f64 (fit, mean, acquisition)" if a.f32 else
                      ("i8 x 6 radix-254 digits, exact products, f64 recombination (variance contraction) / f64 (fit, mean, acquisition)" if a.i8 else "f64")), "data": "synthetic",
            "config": {"workload": "BASELINE configs[" + str(a.config - 1) + "]: m=%d %s-ARD GP, N=%d d=%d, S=%d MC samples, C=%d candidates, noise %g, top-16 selection"
                       % (a.m, a.kernel, a.N, a.d, a.S, a.C, a.noise), "N": a.N, "d": a.d, "m": a.m, "S": a.S, "C": a.C,
                       "parallelism": "candidates sharded over %d GPU(s), %s fit, one all-reduce(MAX) for top-16 (carrier: %s)"
                                      % (world, "output-sharded + broadcast" if a.shard_fit else "replicated", carrier),
                       "rccl_ranks_seen": int(rccl_ranks) if carrier == "native" else (int(dist.get_world_size()) if dist is not None else 0)},
            "value_definition": "K steps with X* and W resident in HBM (the driver's contract: value never includes PCIe); SURVEY 8(d) Metric 1 as "
                                "worded (H2D of X*, D2H of the scores inside every step, median) is the transfers_included block",
            "transfers_included": {"what": "every step also uploads this rank's X* slice (H2D %d B) and downloads its scores (D2H %d B)"
                                           % (Xloc.nbytes, 8 * (hi - lo)),
                                   "value": evals / dt2, "ms_per_step": dt2 / a.steps * 1e3, "ms_per_step_median": float(np.median(per2)) * 1e3},
            "per_rank": None if ranks is None else {
                "ms_per_step": [r[0] / a.steps * 1e3 for r in ranks], "ms_per_step_median": [r[1] * 1e3 for r in ranks],
                "ms_per_step_min_over_ranks": min(r[0] for r in ranks) / a.steps * 1e3, "ms_per_step_max_over_ranks": max(r[0] for r in ranks) / a.steps * 1e3,
                "slowest_single_step_ms": max(r[3] for r in ranks) * 1e3,
                "collective_ms_per_step_rank0": ph2["allreduce"][0] / a.steps,
                "note": "collective = HIP events around ncclAllReduce(MAX) of 2*world*16+1 doubles on rank 0's stream (it includes the wait for the slowest rank); native carrier only"},
            "gp_fit_ms": fit_ms,
            "argmax": int(top_idx[0]),
            "roofline": {"kernel": kernel_name + " (variance contraction V = L^-1 K*, fused column sum-of-squares)",
                         "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TOP/s (int8)" if a.i8 else "TFLOP/s", "frac": ach / peak,
                         "traffic": traffic if not traffic_stale else None,
                         "traffic_source": ("%s (rocprofv3 --pmc passes of this command, kernel %s, taken at git %s)" % (traffic_src, traffic_kernel, traffic_head)
                                            if traffic_src else None),
                         "traffic_stale": traffic_stale,
                         "launch_ms": gemm_ms, "algorithmic_flops_per_launch": gemm_flops,
                         "other_kernels_ms_per_step": {k: v[0] / a.steps for k, v in ph2.items()}},
            "roofline_fit": roofline_fit,
        }
        if i8_block is not None:
            out["int8_variance_contraction"] = i8_block
        if not a.no_cpu_baseline and world == 1:      # CPU baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(p, a, theta)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        lib.bocf_comm_destroy(handle)
        dist.destroy_process_group()


def cpu_baseline(p, a, theta):
    """The oracle (NumPy/SciPy restatement of the reference path, multi-threaded BLAS) timed on this
    host on a bounded sample of the same workload: same fitted model, first `cpu_sample` candidates.  Beside it the
    reference's OWN loop structure -- the interpreted triple loop of uEI_noiseless.py:63-83, which is how the reference
    really evaluates a batch -- timed on a small slice of the same posterior (one core), and the rate measured at survey
    time by running the reference's module itself under the import shim."""
    from oracle import cpu_ref as R          # the only leg of the bench that touches the oracle (besides --check)
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    t0 = time.perf_counter()
    ref = R.MultiOutputGPRef(a.kernel, p["variances"], p["lengthscales"], p["noise"])
    ref.updateModel(p["X"], p["Y"])
    fit_s = time.perf_counter() - t0
    n = min(a.cpu_sample, a.C)
    t0 = time.perf_counter()
    R.batch_uEI(ref, p["Xc"][:n], p["W"], "neg_sq_dist", theta, np.ones(1), "EI")
    dt = time.perf_counter() - t0
    # the reference's loop: for l / for W_s / for candidate i: U(theta, mu_i + sigma_i o W_s), hinge, accumulate
    nl = 32
    mu_eval = ref.posterior_mean_at_evaluated_points()
    mu = ref.posterior_mean(p["Xc"][:nl])
    sigma = np.sqrt(ref.posterior_variance(p["Xc"][:nl]))
    t0 = time.perf_counter()
    R.mc_acq_loop(mu, sigma, mu_eval, p["W"], "neg_sq_dist", theta, np.ones(1), "EI")
    dl = time.perf_counter() - t0
    return {"value": n * a.S / dt, "unit": "evals/s", "cores": int(threads), "kind": "port",
            "sample": "first %d of %d candidates x %d MC samples, same fitted model (N=%d, m=%d); oracle/cpu_ref.py batch_uEI, "
                      "OpenBLAS threads=%d of %d host CPUs" % (n, a.C, a.S, a.N, a.m, threads, os.cpu_count() or 0),
            "seconds": dt, "gp_fit_ms": fit_s * 1e3,
            "as_shipped_loop": {"value": nl * a.S / dl, "unit": "evals/s (MC loop only, posterior given)", "cores": 1,
                                "sample": "%d candidates x %d MC samples through the literal triple loop of uEI_noiseless.py:63-83 "
                                          "(oracle/cpu_ref.py mc_acq_loop), this host" % (nl, a.S), "seconds": dl,
                                "survey_time_reference": {"value": 2.1e5, "unit": "evals/s", "cores": 1,
                                                          "provenance": "BASELINE.md section 2: the reference's own uEI_noiseless.py:63-83 executed "
                                                                        "under the import shim in the survey container (8 vCPU Xeon 2.1 GHz; m=4, 25 W "
                                                                        "samples, 2000 candidates, sequential path), min of 3 runs"}}}


if __name__ == "__main__":
    main()
