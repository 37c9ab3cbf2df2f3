"""Candidate-batch sharding over the GPUs of one node (one process per GPU) and the single
collective the path needs: the global arg-max / top-k of the acquisition values.

The candidate axis is embarrassingly parallel (every candidate's mean, variance and acquisition
value depend only on the fit), so ranks evaluate contiguous slices with NO data-path
communication.  RCCL has no MAXLOC, so the k local winners of every rank are packed into one
float64 buffer of 2*G*k slots (values in [r*k, (r+1)*k), global indices -- exact below 2^53 -- in
[G*k + r*k, ...), -inf elsewhere) and combined with ONE all_reduce(MAX); every rank then merges
the G*k pairs locally (value descending, index ascending = np.argsort(-acq, stable)[:k] of
anchor_points_generator.py:61 on the whole batch).  Payload at G = 8, k = 16: 2 KiB.

Three carriers of that one collective, same packing, same merge, same result:
  * native  -- the context owns an RCCL communicator (`init_native_comm`): local top-k, packing, ncclAllReduce and the merge
               all run on the context's stream inside ONE C call (bocf_global_topk); nothing but the k winners reaches the host;
  * torch   -- torch.distributed with the nccl backend: the library packs into a torch CUDA tensor (bocf_topk_packed), torch
               all-reduces it, the library merges it on the device (bocf_merge_packed);
  * host    -- gloo / CPU ranks (the world_size-2 tests of this container): NumPy packing and merging, below.
"""
import ctypes

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


def _dist(group=None):
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def global_topk(local_idx, local_val, lo, k, group=None, device=None):
    """HOST carrier.  All ranks call this with their local winners (indices relative to their slice start `lo`);
    returns the global (indices, values) of the k best candidates, identical on every rank."""
    dist = _dist(group)
    if dist is None:
        return merge_packed(pack_local_topk(local_idx, local_val, lo, k, 1, 0), k, 1)
    import torch
    G, r = dist.get_world_size(group), dist.get_rank(group)
    buf = pack_local_topk(local_idx, local_val, lo, k, G, r)
    use_cuda = dist.get_backend(group) == "nccl"
    t = torch.from_numpy(buf)
    if use_cuda:
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return merge_packed(t.cpu().numpy(), k, G)


def init_native_comm(model, group=None):
    """Give the model's device context its own RCCL communicator over the ranks of torch.distributed (`group`): rank 0
    creates the rendezvous id, torch.distributed carries the 128 bytes to the other ranks, every rank joins
    (bocf_comm_init is collective).  Returns (world, rank).  torch is only the courier of the id: the collective of the
    path then runs inside the library, on the context's stream."""
    from . import _ffi
    dist = _dist(group)
    lib, ctx = _ffi.load(), model._context()
    if dist is None:
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    have = lib.bocf_comm_info(ctx.handle, None, None)
    if have == 1:
        return world, rank
    ident = ctypes.create_string_buffer(128)
    if rank == 0:
        _ffi.check(lib.bocf_comm_unique_id(ident), "bocf_comm_unique_id")
    if dist is not None and world > 1:
        import torch
        on_gpu = dist.get_backend(group) == "nccl"
        t = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8).clone()
        if on_gpu:
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = ctypes.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
    _ffi.check(lib.bocf_comm_init(ctx.handle, ident, world, rank), "bocf_comm_init")
    return world, rank


def device_global_topk(model, lo, k, group=None):
    """DEVICE carriers.  Global (indices, values) of the k best candidates of the model's last acquisition call, this rank's
    slice starting at global index `lo`; identical on every rank.  Uses the context's own RCCL communicator when it has one
    (`init_native_comm`), else torch.distributed's nccl all-reduce on a buffer the library packs on the device; without any
    process group it is the local selection."""
    from . import _ffi
    lib, ctx = _ffi.load(), model._context()
    idx, val = np.empty(k, dtype=np.int64), np.empty(k)
    pi, pv = idx.ctypes.data_as(_ffi._c_ll_p), _ffi.dptr(val)
    dist = _dist(group)
    native = lib.bocf_comm_info(ctx.handle, None, None) == 1
    if native or dist is None:
        _ffi.check(lib.bocf_global_topk(ctx.handle, k, int(lo), pi, pv), "bocf_global_topk")
    else:
        if dist.get_backend(group) != "nccl":
            raise RuntimeError("device_global_topk needs the nccl backend (RCCL) or a native communicator; CPU ranks use global_topk")
        import torch
        G, r = dist.get_world_size(group), dist.get_rank(group)
        t = torch.empty(2 * G * k, dtype=torch.float64, device=torch.device("cuda", model.device))
        _ffi.check(lib.bocf_topk_packed(ctx.handle, k, int(lo), G, r, ctypes.c_void_p(t.data_ptr())), "bocf_topk_packed")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        torch.cuda.current_stream(t.device).synchronize()       # the merge runs on the library's stream
        _ffi.check(lib.bocf_merge_packed(ctx.handle, k, G, ctypes.c_void_p(t.data_ptr()), pi, pv), "bocf_merge_packed")
    keep = idx >= 0
    return idx[keep], val[keep]


class ShardedBatch(object):
    """Evaluate an acquisition over a candidate batch sharded across the ranks of
    torch.distributed and select the global top-k with one all-reduce."""

    def __init__(self, acquisition, group=None):
        self.acq, self.group = acquisition, group

    def _world(self):
        dist = _dist(self.group)
        if dist is not None:
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def evaluate(self, X, k=16):
        """X: the WHOLE batch (identical on every rank, e.g. drawn from the same seeded RNG as
        GPyOpt/experiment_design/random_design.py:67-77 does).  Returns (local acq (n_local, 1),
        (lo, hi), global top-k indices, values)."""
        G, r = self._world()
        lo, hi = shard_bounds(X.shape[0], G, r)
        a = self.acq._compute_acq(X[lo:hi])
        dist = _dist(self.group)
        model = self.acq.model
        if hasattr(model, "_context") and (dist is None or dist.get_backend(self.group) == "nccl"):
            if hi == lo:
                model._set_candidates(X[lo:hi])              # an empty shard still takes part in the collective
            idx, val = device_global_topk(model, lo, k, self.group)
        else:                                                # CPU ranks (gloo): host carrier
            li = self.acq.select_anchors(min(k, hi - lo)) if hi > lo else np.empty(0, dtype=np.int64)
            idx, val = global_topk(li, a[li, 0], lo, k, self.group)
        return a, (lo, hi), idx, val
