"""Multi-GPU path ON THE GPU (BASELINE configs[3]: candidate shards + one RCCL all-reduce): fresh child processes, one
per GPU, backend nccl -- world_size 1 always, 2 when the box has two devices.  The file sorts first so that its children
are started BEFORE this pytest process touches the GPU (a process that has initialised the GPU must not start programs).
The global top-16 of every carrier must equal the single-process selection over the whole batch bit for bit."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gpu_open_here():
    for fd in os.listdir("/proc/self/fd"):
        try:
            if os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                return True
        except OSError:
            pass
    return False


def _run_ranks(world, tmp_path):
    out = str(tmp_path / ("rccl_w%d.json" % world))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rccl_rank.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode("utf-8", "replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-4000:])
    return json.load(open(out))


_results = {}


def test_children_run_rccl(tmp_path):
    """Phase 1 (no GPU call in this process yet): the ranks."""
    assert not _gpu_open_here(), "this test must run before the pytest process touches the GPU (run the file first / alone)"
    import torch
    n = torch.cuda.device_count()                        # counting devices does not initialise the GPU
    assert n >= 1
    _results[1] = _run_ranks(1, tmp_path)
    if n >= 2:
        _results[2] = _run_ranks(2, tmp_path)


def test_global_topk_equals_single_process():
    """Phase 2: the single-process answer over the whole batch, here, through the same classes."""
    assert _results, "phase 1 did not run"
    import bocf_amd as B
    from bocf_amd.synthetic import synthetic_problem
    N, d, m, C, S = 1024, 6, 4, 8192, 256
    p = synthetic_problem(N, d, m, C, S, 1236)
    kern = [B.kern.RBF(d, variance=p["variances"][j], lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
    model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
    model.updateModel(p["X"], p["Y"])
    theta = np.array([[0.2 * (j + 1) for j in range(m)]])
    U = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), device="neg_sq_dist")
    acq = B.uEI_noiseless(model, None, utility=U)
    acq.W_samples = p["W"]
    a = acq._compute_acq(p["Xc"])[:, 0]
    want = np.argsort(-a, kind="stable")[:16]            # anchor_points_generator.py:61, ties to the lowest index
    idx, val = model.select_topk(16)
    np.testing.assert_array_equal(idx, want)
    a3 = acq._compute_acq(p["Xc"][:3])[:, 0]
    want3 = np.argsort(-a3, kind="stable")
    for world, res in _results.items():
        assert res["world"] == world
        for carrier in ("torch", "native"):
            assert res[carrier]["idx"] == want.tolist(), (world, carrier)
            np.testing.assert_array_equal(np.array(res[carrier]["val"]), a[want])        # bit for bit
        assert res["small"]["idx"] == want3.tolist()
        np.testing.assert_array_equal(np.array(res["small"]["val"]), a3[want3])
        np.testing.assert_array_equal(np.array(res["local_acq_head"]), a[:4])
        # the output-sharded fit over RCCL (one rank: every share is local; two ranks: real broadcasts) == this process's fit
        mean, var = model.predict(p["Xc"][:96])          # the same 96-candidate call as the ranks (<= 16 candidates take the GEMV path)
        np.testing.assert_array_equal(np.array(res["sharded_fit"]["mean"]), mean[:, :8])
        np.testing.assert_array_equal(np.array(res["sharded_fit"]["var"]), var[:, :8])
        np.testing.assert_array_equal(np.array(res["sharded_fit"]["lml"]), model.log_marginal)
