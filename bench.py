"""Benchmark of the hot path: one STEP = one batch acquisition call over the resident candidate
batch = predict (cross kernel K(X,X*), N^2 C variance contraction, mean) + Monte-Carlo uEI +
top-16 selection [+ one all-reduce over ranks].  Default workload = BASELINE.json configs[2]:
m=4 RBF-ARD, N=4096, d=8, S=1024 MC samples, C=65536 candidates, fp64.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0 (metric/value/roofline/cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md, Peak FP32 matrix)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X vendor peak FP64 matrix; v_mfma_f64_16x16x4_f64


def pmc_traffic(N, m, c_local):
    """HBM bytes per launch of the variance GEMM from the committed rocprofv3 PMC passes of THIS command
    (profiles/*/gemm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs; KiB units;
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  None if no matching profile."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for r in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, r, "gemm_traffic.json")
        if os.path.exists(f):
            t = json.load(open(f))
            if (t.get("N"), t.get("m"), t.get("C_local")) == (N, m, c_local):      # later profile directories win
                best = (2.0 * t["FETCH_SIZE_KiB"] + t["WRITE_SIZE_KiB"]) * 1024.0
    return best


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
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 5],
                    help="BASELINE.json config preset (1-based as in SURVEY 8d): 2 = N1024 d6 S256 C8192; 3 = headline; "
                         "5 = m8 Matern52 N8192 d12 S4096 (fp64 here)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="candidates in the CPU-baseline sample")
    ap.add_argument("--f32", action="store_true", help="fp32 variance contraction (option predict_f32; BASELINE configs[4] arithmetic)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="bocf_set_option passthrough (A/B experiments)")
    ap.add_argument("--check", action="store_true", help="parity-check a slice against the oracle before timing")
    return ap.parse_args()


PRESETS = {2: dict(N=1024, d=6, m=4, S=256, C=8192, kernel="rbf", seed=1236),
           5: dict(N=8192, d=12, m=8, S=4096, C=65536, kernel="matern52", seed=1239)}


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
    import torch
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
    from bocf_amd.distributed import global_topk, shard_bounds
    from bocf_amd.synthetic import synthetic_problem

    p = synthetic_problem(a.N, a.d, a.m, a.C, a.S, a.seed)
    kcls = {"rbf": B.kern.RBF, "se": B.kern.SE, "matern52": B.kern.Matern52}[a.kernel]
    kern = [kcls(a.d, variance=p["variances"][j], lengthscale=p["lengthscales"][j], ARD=True) for j in range(a.m)]
    model = B.multi_outputGP(a.m, kernel=kern, noise_var=p["noise"], fixed_hyps=True, device=local_rank)

    if a.f32:
        model.set_option("predict_f32", 1)
    for kv in a.option:
        model.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    # ---- GP fit (metric 2): K build + Cholesky + inverse factor + alpha for all m outputs, incl. H2D
    model.incremental = False                    # time the FULL fit (an unchanged X would otherwise only refresh alpha)
    model.updateModel(p["X"], p["Y"])            # warm-up (allocations)
    fit_ms = []
    for _ in range(3):
        t0 = time.perf_counter()
        model.updateModel(p["X"], p["Y"])
        fit_ms.append((time.perf_counter() - t0) * 1e3)
    fit_ms = float(np.median(fit_ms))

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

    def step():
        model._acq_mc_resident(B._ffi.ACQ_EI, kind, None, theta, np.ones(1), None, fetch=False)
        li, lv = model.select_topk(16)
        return global_topk(li, lv, lo, 16)

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

    for _ in range(a.warmup):
        step()
    model.set_option("profile", 1)
    lib = B._ffi.load()
    import ctypes
    lib.bocf_profile_read(model._context().handle, None, None, None, 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        top_idx, top_val = step()
    fence()
    dt = time.perf_counter() - t0
    ms, launches, flops = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
    lib.bocf_profile_read(model._context().handle, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(flops), 1)
    model.set_option("profile", 0)
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        evals = float(a.C) * a.S * a.steps
        gemm_ms = ms.value / max(1, launches.value)
        gemm_flops = flops.value / max(1, launches.value)          # algorithmic: m N^2 C_local per launch
        ach = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        out = {
            "metric": "acquisition evals/sec (candidates x MC-samples/sec), uEI_noiseless batch call; GP-fit ms alongside",
            "value": evals / dt, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (variance contraction) / f64 (fit, mean, acquisition)" if a.f32 else "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[" + str(a.config - 1) + "]: m=%d %s-ARD GP, N=%d d=%d, S=%d MC samples, C=%d candidates, top-16 selection"
                       % (a.m, a.kernel, a.N, a.d, a.S, a.C), "N": a.N, "d": a.d, "m": a.m, "S": a.S, "C": a.C,
                       "parallelism": "candidates sharded over %d GPU(s), replicated fit, one all-reduce(MAX) for top-16" % world},
            "gp_fit_ms": fit_ms,
            "argmax": int(top_idx[0]),
            "roofline": {"kernel": ("gemm_tn_f32_sumsq_kernel" if a.f32 else ("gemm_tn_f64_sumsq256_kernel" if (hi - lo) >= 32768 and a.N % 256 == 0
                                                                                 and not any(o.startswith("swizzle=") for o in a.option)
                                                                                 else "gemm_tn_f64_kernel<1>")) +
                         " (variance contraction V = L^-1 K*, fused column sum-of-squares)",
                         "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS if a.f32 else FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / (FP32_MFMA_PEAK_TFLOPS if a.f32 else FP64_MFMA_PEAK_TFLOPS),
                         "traffic": None if a.f32 else pmc_traffic(a.N, a.m, hi - lo),
                         "launch_ms": gemm_ms, "algorithmic_flops_per_launch": gemm_flops},
        }
        if not a.no_cpu_baseline and world == 1:      # CPU baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(p, a, theta)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(p, a, theta):
    """The oracle (NumPy/SciPy restatement of the reference path, multi-threaded BLAS) timed on this
    host on a bounded sample of the same workload: same fitted model, first `cpu_sample` candidates."""
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
    return {"value": n * a.S / dt, "unit": "evals/s", "cores": int(threads), "kind": "port",
            "sample": "first %d of %d candidates x %d MC samples, same fitted model (N=%d, m=%d); oracle/cpu_ref.py batch_uEI, "
                      "OpenBLAS threads=%d of %d host CPUs" % (n, a.C, a.S, a.N, a.m, threads, os.cpu_count() or 0),
            "seconds": dt, "gp_fit_ms": fit_s * 1e3}


if __name__ == "__main__":
    main()
