"""Candidate-batch sharding over the GPUs of one node (one process per GPU) and the single
collective the path needs: the global arg-max / top-k of the acquisition values.

The candidate axis is embarrassingly parallel (every candidate's mean, variance and acquisition
value depend only on the replicated fit), so ranks evaluate contiguous slices with NO data-path
communication.  RCCL has no MAXLOC, so the k local winners of every rank are packed into one
float64 buffer of 2*G*k slots (values in [r*k, (r+1)*k), global indices -- exact below 2^53 -- in
[G*k + r*k, ...), -inf elsewhere) and combined with ONE all_reduce(MAX); every rank then merges
the G*k pairs locally (value descending, index ascending = np.argsort(-acq, stable)[:k] of
anchor_points_generator.py:61 on the whole batch).  Payload at G = 8, k = 16: 2 KiB.
"""
import numpy as np


def shard_bounds(C, world_size, rank):
    """Contiguous slice [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(C, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_local_topk(local_idx, local_val, lo, k, world_size, rank):
    buf = np.full(2 * world_size * k, -np.inf)
    n = len(local_idx)
    buf[rank * k: rank * k + n] = local_val
    buf[world_size * k + rank * k: world_size * k + rank * k + n] = np.asarray(local_idx, dtype=np.float64) + lo
    return buf


def merge_packed(buf, k, world_size):
    vals, idx = buf[: world_size * k], buf[world_size * k:]
    keep = np.isfinite(idx)
    vals, idx = vals[keep], idx[keep].astype(np.int64)
    order = np.lexsort((idx, -vals))[:k]
    return idx[order], vals[order]


def global_topk(local_idx, local_val, lo, k, group=None, device=None):
    """All ranks call this with their local winners (indices relative to their slice start `lo`);
    returns the global (indices, values) of the k best candidates, identical on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return merge_packed(pack_local_topk(local_idx, local_val, lo, k, 1, 0), k, 1)
    G, r = dist.get_world_size(group), dist.get_rank(group)
    buf = pack_local_topk(local_idx, local_val, lo, k, G, r)
    use_cuda = dist.get_backend(group) == "nccl"
    t = torch.from_numpy(buf)
    if use_cuda:
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return merge_packed(t.cpu().numpy(), k, G)


class ShardedBatch(object):
    """Evaluate an acquisition over a candidate batch sharded across the ranks of
    torch.distributed and select the global top-k with one all-reduce."""

    def __init__(self, acquisition, group=None):
        self.acq, self.group = acquisition, group

    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def evaluate(self, X, k=16):
        """X: the WHOLE batch (identical on every rank, e.g. drawn from the same seeded RNG as
        GPyOpt/experiment_design/random_design.py:67-77 does).  Returns (local acq (n_local, 1),
        (lo, hi), global top-k indices, values)."""
        G, r = self._world()
        lo, hi = shard_bounds(X.shape[0], G, r)
        a = self.acq._compute_acq(X[lo:hi])
        li = self.acq.select_anchors(min(k, hi - lo)) if hi > lo else np.empty(0, dtype=np.int64)
        idx, val = global_topk(li, a[li, 0], lo, k, self.group)
        return a, (lo, hi), idx, val
