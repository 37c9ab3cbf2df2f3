"""One rank of the RCCL test (started by tests/test_00_gpu_rccl.py as a FRESH process, one per GPU).  Not collected by pytest.

Fits the config-2 model, scores this rank's contiguous slice of the 8192-candidate batch and selects the global top-16
through every device carrier of the one collective (bocf_amd/distributed.py): the context's own RCCL communicator
(bocf_global_topk) and torch.distributed's nccl all-reduce over a buffer packed on the device.  Rank 0 writes the result."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    import numpy as np
    import torch                                         # before libbocf_hip.so: both then share ONE HIP runtime (libamdhip64.so.7)
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    import bocf_amd as B
    from bocf_amd.distributed import ShardedBatch, device_global_topk, init_native_comm, shard_bounds
    from bocf_amd.synthetic import synthetic_problem
    N, d, m, C, S = 1024, 6, 4, 8192, 256                # BASELINE configs[1]
    p = synthetic_problem(N, d, m, C, S, 1236)
    kern = [B.kern.RBF(d, variance=p["variances"][j], lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
    model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True, device=local)
    model.updateModel(p["X"], p["Y"])
    theta = np.array([[0.2 * (j + 1) for j in range(m)]])
    U = B.Utility(parameter_dist=B.ParameterDistribution(support=theta, prob_dist=np.ones(1)), device="neg_sq_dist")
    acq = B.uEI_noiseless(model, None, utility=U)
    acq.W_samples = p["W"]
    res = {"world": world}
    # carrier 1: torch.distributed (nccl = RCCL) all-reduce over the device-packed buffer
    sb = ShardedBatch(acq)
    a, (lo, hi), idx, val = sb.evaluate(p["Xc"], k=16)
    assert (lo, hi) == shard_bounds(C, world, rank)
    res["torch"] = {"idx": idx.tolist(), "val": val.tolist()}
    # carrier 2: the context's own communicator, one C call
    w, r = init_native_comm(model)
    lib = B._ffi.load()
    import ctypes
    cw, cr = ctypes.c_int(), ctypes.c_int()
    assert lib.bocf_comm_info(model._context().handle, ctypes.byref(cw), ctypes.byref(cr)) == 1
    assert (cw.value, cr.value) == (world, rank) == (w, r)
    idx2, val2 = device_global_topk(model, lo, 16)
    res["native"] = {"idx": idx2.tolist(), "val": val2.tolist()}
    # an empty shard still takes part: 3 candidates over `world` ranks, k = 16 -> the 3 candidates, sorted
    small = p["Xc"][:3]
    _, _, idx3, val3 = ShardedBatch(acq).evaluate(small, k=16)
    res["small"] = {"idx": idx3.tolist(), "val": val3.tolist()}
    # output-sharded fit over the same communicator: rank r factorizes its share of the 4 outputs, the inverse factors travel
    # by RCCL broadcast, the small vectors by one all-reduce; predictions must equal the replicated fit's bit for bit
    mean_rep, var_rep = model.predict(p["Xc"][:96])
    model.set_option("shard_fit", 1)
    model.updateModel(p["X"], p["Y"])
    mean_sh, var_sh = model.predict(p["Xc"][:96])
    assert np.array_equal(mean_sh, mean_rep) and np.array_equal(var_sh, var_rep), "sharded fit differs from the replicated fit"
    res["sharded_fit"] = {"mean": mean_sh[:, :8].tolist(), "var": var_sh[:, :8].tolist(), "lml": model.log_marginal.tolist()}
    model.set_option("shard_fit", 0)
    # every rank must hold the same answer
    t = torch.tensor(idx2.tolist() + idx.tolist(), device="cuda", dtype=torch.float64)
    t0 = t.clone()
    dist.broadcast(t0, src=0)
    assert torch.equal(t, t0), "ranks disagree on the global top-k"
    res["local_acq_head"] = a[:4, 0].tolist()
    dist.barrier()
    if rank == 0:
        with open(out, "w") as f:
            json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
