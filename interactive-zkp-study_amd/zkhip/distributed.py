"""Multi-GPU MSM: shard by contiguous point chunks, one process per GPU, one tiny all-gather.

MSM is linear, so rank g computes the full Pippenger pipeline on points [lo_g, hi_g) down to a
single projective partial sum (128 bytes for G1).  The partials are exchanged with ONE
all-gather (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests) and folded in rank
order on the host -- RCCL has no elliptic-curve reduction operator, so the "all-reduce of partial
sums" is all-gather + deterministic local fold.  The message is latency-bound (a few hundred
bytes), so xGMI bandwidth is irrelevant; no bucket data ever crosses the fabric.
"""
import ctypes

import numpy as np

from . import _lib
from .field import limbs_to_g1, limbs_to_g2


def shard_range(n, rank, world_size):
    """Contiguous chunk [lo, hi) of rank `rank`; chunk sizes differ by at most one."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


_gather_ctx = {}


def all_gather_partials(partial, device=None, group=None):
    """partial: uint64[L] on the host -> uint64[world, L] (rank order) on every rank.

    On a GPU the copy-in, the collective and the copy-out run on a side stream of their own, so they wait
    for nothing the caller has queued on its compute stream since (bench.py keeps the next MSM in flight
    while the previous step's partials are exchanged); buffers are allocated once per (device, size)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    src = np.ascontiguousarray(partial).view(np.int64)
    if device is None or torch.device(device).type != "cuda":
        t = torch.from_numpy(src.copy())
        out = torch.empty(world * t.numel(), dtype=torch.int64)  # flat: gloo and RCCL both accept it
        dist.all_gather_into_tensor(out, t, group=group)
        return out.numpy().view(np.uint64).reshape(world, -1)
    key = (str(device), src.size, world)
    ctx = _gather_ctx.get(key)
    if ctx is None:
        ctx = {"stream": torch.cuda.Stream(device=device, priority=-1),  # ahead of the bulk MSM kernels
               "h_in": torch.empty(src.size, dtype=torch.int64).pin_memory(),
               "d_in": torch.empty(src.size, dtype=torch.int64, device=device),
               "d_out": torch.empty(world * src.size, dtype=torch.int64, device=device),
               "h_out": torch.empty(world * src.size, dtype=torch.int64).pin_memory()}
        _gather_ctx[key] = ctx
    ctx["h_in"].numpy()[:] = src
    with torch.cuda.stream(ctx["stream"]):
        ctx["d_in"].copy_(ctx["h_in"], non_blocking=True)
        dist.all_gather_into_tensor(ctx["d_out"], ctx["d_in"], group=group)
        ctx["h_out"].copy_(ctx["d_out"], non_blocking=True)
    ctx["stream"].synchronize()
    return ctx["h_out"].numpy().view(np.uint64).reshape(world, -1).copy()


def fold_partials(group_id, partials):
    """Sum of the partial sums (rank order) -> affine point or None (zk_msm_fold_partials)."""
    partials = np.ascontiguousarray(partials, dtype=np.uint64)
    limbs = 8 if group_id == _lib.GROUP_G1 else 16
    count = partials.size // (2 * limbs)
    out = np.zeros(limbs, dtype=np.uint64)
    inf = ctypes.c_int(0)
    _lib.check(_lib.load().zk_msm_fold_partials(group_id, _lib.ptr(partials), count, _lib.ptr(out), ctypes.byref(inf)))
    if inf.value:
        return None
    return (limbs_to_g1(out) if group_id == _lib.GROUP_G1 else limbs_to_g2(out))[0]


def sharded_msm(group_id, local_partial, device=None, group=None):
    """local_partial: this rank's XYZZ partial (uint64[16|32]) -> the global MSM result on every rank."""
    return fold_partials(group_id, all_gather_partials(local_partial, device=device, group=group))
