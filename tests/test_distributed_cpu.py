"""N>1 path on CPU: candidate sharding + the single all_reduce(MAX) for the global top-k, with the
gloo backend and world_size 2 (and the pure merge logic for more ranks)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_cover_and_balance():
    from bocf_amd.distributed import shard_bounds
    for C in (0, 1, 7, 16, 65536, 65537):
        for G in (1, 2, 3, 8):
            b = [shard_bounds(C, G, r) for r in range(G)]
            assert b[0][0] == 0 and b[-1][1] == C
            assert all(b[i][1] == b[i + 1][0] for i in range(G - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("G", [1, 2, 4, 8])
def test_pack_merge_equals_global_argsort(G):
    from bocf_amd.distributed import merge_packed, pack_local_topk, shard_bounds
    rng = np.random.RandomState(G)
    C, k = 1000, 16
    acq = np.round(rng.rand(C), 2)                       # many exact ties
    acq[rng.randint(0, C, 400)] = 0.0
    bufs = []
    for r in range(G):
        lo, hi = shard_bounds(C, G, r)
        li = np.argsort(-acq[lo:hi], kind="stable")[:k]
        bufs.append(pack_local_topk(li, acq[lo:hi][li], lo, k, G, r))
    red = np.max(np.stack(bufs), axis=0)                  # what all_reduce(MAX) produces
    idx, val = merge_packed(red, k, G)
    want = np.argsort(-acq, kind="stable")[:k]
    np.testing.assert_array_equal(idx, want)
    np.testing.assert_array_equal(val, acq[want])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from bocf_amd.distributed import global_topk, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(0)                        # every rank draws the same batch
    acq = rng.rand(4097)
    acq[[5, 2050, 4000]] = 2.0                            # exact ties across shards -> lowest index first
    lo, hi = shard_bounds(acq.size, world, rank)
    li = np.argsort(-acq[lo:hi], kind="stable")[:16]
    idx, val = global_topk(li, acq[lo:hi][li], lo, 16)
    q.put((rank, idx.tolist(), val.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_global_topk_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.RandomState(0)
    acq = rng.rand(4097)
    acq[[5, 2050, 4000]] = 2.0
    want = np.argsort(-acq, kind="stable")[:16]
    for _, idx, val in res:
        assert idx == want.tolist()
        assert idx[:3] == [5, 2050, 4000]
        np.testing.assert_array_equal(val, acq[want])


def _worker_small(rank, world, port, q, C):
    """Fewer candidates than ranks x k: some shards hold less than k candidates, one may be EMPTY; the packed buffer pads with
    (-inf, -1) and every rank still ends with the same global selection."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from bocf_amd.distributed import global_topk, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acq = np.random.RandomState(3).rand(C)
    lo, hi = shard_bounds(C, world, rank)
    li = np.argsort(-acq[lo:hi], kind="stable")[:16]
    idx, val = global_topk(li, acq[lo:hi][li], lo, 16)
    q.put((rank, idx.tolist(), val.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,C", [(3, 10), (4, 3)])
def test_global_topk_gloo_uneven_and_empty_shards(world, C):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_small, args=(r, world, port, q, C)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    acq = np.random.RandomState(3).rand(C)
    want = np.argsort(-acq, kind="stable")[:16]
    for _, idx, val in res:
        assert idx[:C] == want.tolist()                   # all C candidates, best first ...
        assert all(i == -1 for i in idx[C:])              # ... then empty slots (index -1, value -inf)
        np.testing.assert_array_equal(val[:C], acq[want])
        assert all(v == -np.inf for v in val[C:])
