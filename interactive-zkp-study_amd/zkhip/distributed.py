"""Multi-GPU MSM (shard by contiguous point chunks, one tiny all-gather) and the multi-GPU single large NTT
(four-step, one all-to-all); one process per GPU.

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
_RING = 6   # exchanges that may be in flight per (device, size, world)


class PartialGather:
    """One all-gather of a host partial, started by all_gather_partials_start and finished by .result().

    On a GPU the copy-in, the collective (RCCL) and the copy-out are only ENQUEUED by the start call -- on a side stream of
    their own, so they wait for nothing the caller has queued on its compute stream since -- and .result() waits for the
    copy-out.  A caller that keeps MSMs in flight starts the exchange of step k and collects the one of step k-1: the
    collective's kernel has to find its way onto CUs that the MSM kernels keep full, and that latency (0.2 ms per step on one
    rank when waited for at once) then hides behind the next MSM instead of stalling the submission loop."""

    def __init__(self, slot=None, event=None, value=None):
        self._slot, self._event, self._value = slot, event, value

    def done(self):
        """True when .result() would not block (the copy-out has landed)."""
        return self._value is not None or self._event.query()

    def result(self):
        """-> uint64[world, L] (rank order)."""
        if self._value is None:
            self._event.synchronize()
            self._value = self._slot["h_out"].numpy().view(np.uint64).reshape(self._slot["world"], -1).copy()
            self._slot["busy"] = False
        return self._value

    def abandon(self):
        """Gives the ring slot back without reading the result (error paths): waits for whatever was enqueued on it."""
        if self._value is None and self._slot is not None:
            try:
                self._event.synchronize()
            finally:
                self._slot["busy"] = False


def all_gather_partials_start(partial, device=None, group=None):
    """partial: uint64[L] on the host -> PartialGather.  Buffers are allocated once per (device, size, world), as a ring."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    src = np.ascontiguousarray(partial).view(np.int64)
    if device is None or torch.device(device).type != "cuda":
        t = torch.from_numpy(src.copy())
        out = torch.empty(world * t.numel(), dtype=torch.int64)  # flat: gloo and RCCL both accept it
        dist.all_gather_into_tensor(out, t, group=group)
        return PartialGather(value=out.numpy().view(np.uint64).reshape(world, -1))
    key = (str(device), src.size, world)
    ctx = _gather_ctx.get(key)
    if ctx is None:
        ctx = {"stream": torch.cuda.Stream(device=device, priority=-1), "next": 0, "slots": []}  # ahead of the bulk MSM kernels
        for _ in range(_RING):
            ctx["slots"].append({"h_in": torch.empty(src.size, dtype=torch.int64).pin_memory(),
                                 "d_in": torch.empty(src.size, dtype=torch.int64, device=device),
                                 "d_out": torch.empty(world * src.size, dtype=torch.int64, device=device),
                                 "h_out": torch.empty(world * src.size, dtype=torch.int64).pin_memory(),
                                 "event": torch.cuda.Event(), "busy": False, "world": world})
        _gather_ctx[key] = ctx
    slot = ctx["slots"][ctx["next"]]
    if slot["busy"]:                      # checked BEFORE the ring advances: a refused start leaves the ring where it was
        raise RuntimeError("all_gather_partials_start: more than %d exchanges in flight; collect the oldest first" % _RING)
    ctx["next"] = (ctx["next"] + 1) % _RING
    slot["busy"] = True
    slot["h_in"].numpy()[:] = src
    with torch.cuda.stream(ctx["stream"]):
        slot["d_in"].copy_(slot["h_in"], non_blocking=True)
        dist.all_gather_into_tensor(slot["d_out"], slot["d_in"], group=group)   # enqueued; the host does not wait for it
        slot["h_out"].copy_(slot["d_out"], non_blocking=True)
        slot["event"].record(ctx["stream"])
    return PartialGather(slot=slot, event=slot["event"])


def all_gather_partials(partial, device=None, group=None):
    """partial: uint64[L] on the host -> uint64[world, L] (rank order) on every rank (start + wait)."""
    return all_gather_partials_start(partial, device=device, group=group).result()


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


class ShardedMsmResult:
    """Handle of a sharded MSM whose partial exchange is in flight (sharded_msm_start); .result() -> the global point."""

    def __init__(self, group_id, gather):
        self._group_id, self._gather = group_id, gather

    def done(self):
        return self._gather.done()

    def result(self):
        return fold_partials(self._group_id, self._gather.result())

    def abandon(self):
        self._gather.abandon()


def sharded_msm_start(group_id, local_partial, device=None, group=None):
    """The pipelined form of sharded_msm: starts the exchange of this rank's partial and returns at once."""
    return ShardedMsmResult(group_id, all_gather_partials_start(local_partial, device=device, group=group))


class ExchangeWorker:
    """The exchange side of a pipelined multi-rank MSM loop, on a thread of its own: post() hands over a collected partial; the
    thread starts its all-gather (sharded_msm_start), finishes the exchanges in order -- as soon as one has landed, at the latest
    three posts later -- and folds them.  That is ~0.1 ms of Python, collective launch and host arithmetic per step, which
    otherwise sits between a lane finishing and its next submission; here it runs while the submitting thread blocks inside
    zk_msm_collect (ctypes releases the GIL).  All collectives of the loop are issued by this one thread, in post order (the same
    on every rank); flush() returns the last folded result once every exchange handed over has been folded, so nothing is in
    flight when the caller goes on to a barrier.

    Errors: an exception in the thread is kept and re-raised by the next flush(); the exchanges in flight at that moment are
    waited for and their ring slots released, and the posts that follow until that flush() are NOT exchanged -- the other ranks
    of a collective that this rank no longer joins would block, so a caller treats an error from flush() as fatal for the
    process group (bench.py exits).  close() ends the thread."""

    def __init__(self, group_id, device=None, group=None, cuda_device=None, lag=3):
        import queue
        import threading
        self._args = (group_id, device, group, cuda_device, lag)
        self.q, self.idle, self.res, self.err = queue.Queue(), threading.Event(), None, None
        self.idle.set()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    _STOP = object()

    def _run(self):
        group_id, device, group, cuda_device, lag = self._args
        if cuda_device is not None:
            import torch
            torch.cuda.set_device(cuda_device)      # the current device is per thread
        inflight = []
        while True:
            item = self.q.get()
            try:
                if item is None or item is self._STOP:   # flush / close: finish everything, in order
                    while inflight:
                        self.res = inflight.pop(0).result()
                    self.idle.set()
                    if item is self._STOP:
                        return
                    continue
                if self.err is None:
                    inflight.append(sharded_msm_start(group_id, item, device=device, group=group))
                    while inflight and (len(inflight) > lag or inflight[0].done()):
                        self.res = inflight.pop(0).result()
            except BaseException as exc:             # noqa: BLE001 -- handed to the submitting thread by flush()
                self.err = exc
                for handle in inflight:              # give their ring slots back (waits for what was enqueued)
                    try:
                        handle.abandon()
                    except BaseException:            # noqa: BLE001
                        pass
                inflight.clear()
                if item is None or item is self._STOP:
                    self.idle.set()
                if item is self._STOP:
                    return

    def post(self, partial):
        self.idle.clear()
        self.q.put(partial)

    def flush(self):
        self.idle.clear()
        self.q.put(None)
        self.idle.wait()
        if self.err is not None:
            err, self.err = self.err, None
            raise err
        return self.res

    def close(self):
        """Finishes what is in flight and ends the thread (idempotent)."""
        if self._thread is not None and self._thread.is_alive():
            self.idle.clear()
            self.q.put(self._STOP)
            self._thread.join(timeout=60)
        self._thread = None


# ------------------------------------------------------------------------------------------------
# Single large NTT across GPUs (mode 2): four-step with ONE all-to-all.
#
# n = n1 * n2 points, index j = j1 * n2 + j2 in, k = k1 + n1 * k2 out:
#     X[k1 + n1 k2] = sum_j2 w_n2^(j2 k2) * [ w_n^(j2 k1) * sum_j1 x[j1 n2 + j2] w_n1^(j1 k1) ]
# Layout "block-cyclic BC(m)": rank r owns the elements whose index i satisfies (i mod m) in [r m/R, (r+1) m/R),
# stored residue-major as the matrix [(i mod m) - r m/R][i div m] (each owned residue class is one contiguous row).  The
# forward transform takes BC(n2) and leaves BC(n1) (the same layout when n1 == n2, i.e. for even log n); the inverse
# takes BC(n1) back to BC(n2).  Pointwise work between a forward and an inverse transform (the quotient of a prover)
# is layout-agnostic, and an MSM over BC-distributed scalars only needs its points distributed the same way.
#   1. local: the n2/R owned columns are contiguous rows -> n2/R transforms of length n1; the last pass multiplies by the twiddle
#      w_n^(j2 k1) and stores every result straight into its destination's block of the send buffer
#   2. ONE all-to-all: rank r sends rank s the k1-block of s of its columns (n/R^2 elements per pair; every pair of GPUs has its
#      own xGMI link, so all links carry traffic at once)
#   3. local: n1/R transforms of length n2 whose first pass reads the received [j2][k1 local] matrix transposed -- already the
#      BC(n1) storage.
# On the GPU steps 1 and 3 are zk_ntt_dev_io calls (the pack, the twiddle and the transpose happen in the pass kernels' first
# loads / last stores); `_HipLocal.fused` marks that.  The CPU tests inject a local object without it and take the explicit
# ntt_rows / twiddle / permute route below, which defines what the fused kernels must reproduce.
class _HipLocal:
    """The local kernels of DistNtt on this process's GPU."""
    fused = True

    def __init__(self, log_n, l1, l2):
        from .device import NttPlan
        self.p1, self.p2 = NttPlan(l1), NttPlan(l2)
        self.pn = NttPlan(log_n)  # twiddle tables of the full size (its scratch is never allocated)

    def ntt_rows(self, t, which, inverse):
        """t: (batch, len, 4) int64 contiguous device tensor, transformed in place along dim 1."""
        import torch
        plan = self.p1 if which == 1 else self.p2
        plan.run_batch(t.data_ptr(), t.shape[0], inverse, torch.cuda.current_stream().cuda_stream)

    def twiddle(self, t, row0, inverse):
        """t: (rows, cols, 4): t[b, k] *= w_n^(+-(row0 + b) k)."""
        import torch
        self.pn.twiddle(t.data_ptr(), t.shape[1].bit_length() - 1, t.shape[0], row0, inverse, torch.cuda.current_stream().cuda_stream)

    def io(self, which, src, dst, batch, inverse, in_layout, out_layout, log_block=0, row0=0):
        """zk_ntt_dev_io on the current stream: `batch` transforms of dimension `which` from src to dst in the given layouts."""
        import torch
        plan = self.p1 if which == 1 else self.p2
        plan.run_io(src.data_ptr(), dst.data_ptr(), batch, inverse, in_layout, out_layout, log_block, row0, self.pn, inverse,
                    torch.cuda.current_stream().cuda_stream)


class DistNtt:
    """Forward / inverse NTT of 2^log_n points spread over the ranks of a process group (see the comment above).

    `local` supplies ntt_rows / twiddle; the default runs the HIP kernels (tests inject the oracle to
    exercise the layout logic and the collective on CPU)."""

    def __init__(self, log_n, group=None, l1=None, local=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.log_n = int(log_n)
        self.l1 = int(l1) if l1 is not None else self.log_n // 2
        if not 0 <= self.l1 <= self.log_n:
            raise ValueError("DistNtt: l1 must lie in [0, log_n]")
        self.l2 = self.log_n - self.l1
        self.n1, self.n2 = 1 << self.l1, 1 << self.l2
        R = self.world
        if R & (R - 1) or self.n1 % R or self.n2 % R:
            raise ValueError("DistNtt: the world size must be a power of two dividing both 2^l1 and 2^l2")
        self.c, self.k = self.n2 // R, self.n1 // R  # local columns (forward input) / local rows (forward output)
        self.local = local if local is not None else _HipLocal(self.log_n, self.l1, self.l2)

    # layouts ------------------------------------------------------------------------------------
    def local_shape_in(self):
        return (self.c, self.n1)      # BC(n2): [j2 local][j1]

    def local_shape_out(self):
        return (self.k, self.n2)      # BC(n1): [k1 local][k2]

    def scatter_in(self, full):
        """Natural-order array (n, 4) -> this rank's BC(n2) block (c, n1, 4) (test / setup helper)."""
        return np.ascontiguousarray(full.reshape(self.n1, self.n2, 4)[:, self.rank * self.c:(self.rank + 1) * self.c].transpose(1, 0, 2))

    def scatter_out(self, full):
        return np.ascontiguousarray(full.reshape(self.n2, self.n1, 4)[:, self.rank * self.k:(self.rank + 1) * self.k].transpose(1, 0, 2))

    # transforms ---------------------------------------------------------------------------------
    def _exchange(self, send):
        """send: (R, a, b, 4) contiguous, block s goes to rank s -> (R, a, b, 4) with block r received from rank r."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return send
        if send.is_cuda and dist.get_backend(self.group) == "gloo":   # rehearsal of several ranks on one GPU: stage through the host
            h_send = send.cpu()
            h_recv = torch.empty_like(h_send)
            dist.all_to_all_single(h_recv.view(-1), h_send.view(-1), group=self.group)
            return h_recv.to(send.device)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group)
        return recv

    def _exchange_start(self, send):
        """_exchange without waiting: -> (recv, work).  `work` is None when nothing is outstanding (one rank, or the host-staged
        rehearsal of several ranks on one GPU); otherwise work.wait() orders the CURRENT stream behind the all-to-all, which RCCL
        runs on a stream of its own -- kernels enqueued between the two calls run beside the exchange."""
        import torch
        import torch.distributed as dist
        if self.world == 1 or (send.is_cuda and dist.get_backend(self.group) == "gloo"):
            return self._exchange(send), None
        recv = torch.empty_like(send)
        return recv, dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group, async_op=True)

    def forward_many(self, xs):
        """forward() of several vectors in the block-cyclic layouts, the exchanges in flight together: vector j's all-to-all runs
        under vector j+1's first local transform, and every second local transform starts when ITS exchange has landed (a
        prover's A, B, C: three exchanges, two of them hidden).  Same results as forward() one by one."""
        R, n1, n2, c, k = self.world, self.n1, self.n2, self.c, self.k
        fused = getattr(self.local, "fused", False)
        pending = []
        for x in xs:
            if fused:
                import torch
                from ._lib import NTT_BLOCKED_TW, NTT_PLAIN
                send = torch.empty((R, c, k, 4), dtype=x.dtype, device=x.device)
                self.local.io(1, x, send, c, False, NTT_PLAIN, NTT_BLOCKED_TW, k.bit_length() - 1, self.rank * c)
            else:
                self.local.ntt_rows(x, 1, False)
                self.local.twiddle(x, self.rank * c, False)
                send = x.view(c, R, k, 4).permute(1, 0, 2, 3).contiguous()
            pending.append((x, send) + self._exchange_start(send))
        out = []
        for x, send, recv, work in pending:
            if work is not None:
                work.wait()
            if fused:
                import torch
                from ._lib import NTT_PLAIN, NTT_TRANSPOSED
                rows = x.view(-1)[:k * n2 * 4].view(k, n2, 4) if x.numel() == k * n2 * 4 else torch.empty((k, n2, 4), dtype=x.dtype, device=x.device)
                self.local.io(2, recv, rows, k, False, NTT_TRANSPOSED, NTT_PLAIN)
            else:
                rows = recv.view(n2, k, 4).permute(1, 0, 2).contiguous()
                self.local.ntt_rows(rows, 2, False)
            out.append(rows)
        return out

    def inverse_many(self, ys):
        """inverse() of several vectors in the block-cyclic layouts with their exchanges in flight together (see forward_many)."""
        R, n1, n2, c, k = self.world, self.n1, self.n2, self.c, self.k
        fused = getattr(self.local, "fused", False)
        pending = []
        for y in ys:
            if fused:
                import torch
                from ._lib import NTT_PLAIN, NTT_TRANSPOSED
                send = torch.empty((R, c, k, 4), dtype=y.dtype, device=y.device)
                self.local.io(2, y, send, k, True, NTT_PLAIN, NTT_TRANSPOSED)
            else:
                self.local.ntt_rows(y, 2, True)
                send = y.view(k, R, c, 4).permute(1, 2, 0, 3).contiguous()
            pending.append((y, send) + self._exchange_start(send))
        out = []
        for y, send, recv, work in pending:
            if work is not None:
                work.wait()
            if fused:
                from ._lib import NTT_BLOCKED_TW, NTT_PLAIN
                cols = y.view(-1)[:c * n1 * 4].view(c, n1, 4)
                self.local.io(1, recv, cols, c, True, NTT_BLOCKED_TW, NTT_PLAIN, k.bit_length() - 1, self.rank * c)
            else:
                cols = recv.permute(1, 0, 2, 3).contiguous().view(c, n1, 4)
                self.local.twiddle(cols, self.rank * c, True)
                self.local.ntt_rows(cols, 1, True)
            out.append(cols)
        return out

    # natural order across ranks ------------------------------------------------------------------
    # The block-cyclic layouts are what a prover's chain needs (transform -> pointwise -> inverse -> MSM never leaves them).  Where
    # the NATURAL order is wanted across ranks -- rank r holding the contiguous slice [r n/R, (r+1) n/R) -- it is one more
    # all-to-all of n/R^2 elements per pair and a local transpose ("output order must be fixed up to
    # natural order"): BC(n1) stores X[k1 + n1 k2] at [k1 local][k2]; the natural slice of rank s is k2 in its n2/R block, all k1.
    # These four convert a vector that already sits in a block-cyclic layout; forward / inverse (natural_in, natural_out) go from
    # and to the natural order directly, with the transposes folded into the pass kernels' loads and stores.
    def bc_out_to_natural(self, y):
        """(k, n2, 4) BC(n1) block (what forward() returns) -> (n/R, 4): this rank's contiguous slice of the natural order."""
        R, k, c, n1 = self.world, self.k, self.c, self.n1
        send = y.view(k, R, c, 4).permute(1, 0, 2, 3).contiguous()      # [s][k1 local][k2 local of s]
        recv = self._exchange(send)                                     # [r][k1 local of r][k2 local]  ==  [k1][k2 local]
        return recv.view(n1, c, 4).permute(1, 0, 2).contiguous().view(c * n1, 4)   # [k2 local][k1]: index k1 + n1 k2, k2 in this rank's block

    def natural_to_bc_out(self, z):
        """The inverse of bc_out_to_natural: (n/R, 4) contiguous slice -> (k, n2, 4) BC(n1) block (what inverse() takes)."""
        R, k, c, n1, n2 = self.world, self.k, self.c, self.n1, self.n2
        send = z.view(c, R, k, 4).permute(1, 2, 0, 3).contiguous()      # [s][k1 local of s][k2 local]
        recv = self._exchange(send)                                     # [r][k1 local][k2 local of r]
        return recv.permute(1, 0, 2, 3).contiguous().view(k, n2, 4)     # [k1 local][k2]

    def natural_to_bc_in(self, z):
        """(n/R, 4) contiguous slice of a natural-order vector x[j1 n2 + j2] -> (c, n1, 4) BC(n2) block (what forward() takes)."""
        R, k, c, n1, n2 = self.world, self.k, self.c, self.n1, self.n2
        send = z.view(k, R, c, 4).permute(1, 0, 2, 3).contiguous()      # rows j1 in this rank's block: [s][j1 local][j2 local of s]
        recv = self._exchange(send)                                     # [r][j1 local of r][j2 local]  ==  [j1][j2 local]
        return recv.view(n1, c, 4).permute(1, 0, 2).contiguous()        # [j2 local][j1]

    def bc_in_to_natural(self, x):
        """The inverse of natural_to_bc_in: (c, n1, 4) BC(n2) block (what inverse() returns) -> (n/R, 4) contiguous slice."""
        R, k, c, n2 = self.world, self.k, self.c, self.n2
        send = x.view(c, R, k, 4).permute(1, 2, 0, 3).contiguous()      # [s][j1 local of s][j2 local]
        recv = self._exchange(send)                                     # [r][j1 local][j2 local of r]
        return recv.permute(1, 0, 2, 3).contiguous().view(k * n2, 4)    # [j1 local][j2]: index j1 n2 + j2

    def forward(self, x, natural_in=False, natural_out=False):
        """x: (c, n1, 4) int64 tensor, this rank's BC(n2) block (transformed in place as scratch) -> (k, n2, 4), its
        BC(n1) block of the transform.
        natural_in: x is the rank's contiguous slice x[r n/R .. (r+1) n/R) of the natural order instead (any shape with n/R
        elements); natural_out: the result is the rank's contiguous slice of X, (n/R, 4).  Across ranks each of the two costs
        one more all-to-all: the index the LAST local transform runs over is the high part of the output index
        (k = k1 + n1 k2), so the rank that holds a k1 holds a strided set whatever the factorisation -- a contiguous slice
        needs a second exchange.  What round 5 removed are the passes around it: the transposes happen in the pass kernels'
        first loads / last stores (ZK_NTT_TRANSPOSED: the stores of the last pass ARE the send buffer, block s of it bound for
        rank s), and one strided copy orders the received blocks.  On one rank nothing is exchanged."""
        R, n1, n2, c, k = self.world, self.n1, self.n2, self.c, self.k
        fused = getattr(self.local, "fused", False)
        if (natural_in or natural_out) and R == 1 and not fused:
            raise ValueError("DistNtt: on one rank natural-order input / output is offered with the GPU kernels only")
        if natural_in and R > 1:
            send = x.reshape(k, R, c, 4).permute(1, 0, 2, 3).contiguous()   # rows j1 of this rank's block: [s][j1 local][j2 local of s]
            x = self._exchange(send)                                        # [r][j1 local of r][j2 local]  ==  [j1][j2 local]: transposed BC(n2)
            if not fused:
                x = x.view(n1, c, 4).permute(1, 0, 2).contiguous()          # [j2 local][j1]
        if fused:
            import torch
            from ._lib import NTT_BLOCKED_TW, NTT_PLAIN, NTT_TRANSPOSED
            send = torch.empty((R, c, k, 4), dtype=x.dtype, device=x.device)
            self.local.io(1, x, send, c, False, NTT_TRANSPOSED if natural_in else NTT_PLAIN, NTT_BLOCKED_TW, k.bit_length() - 1, self.rank * c)
            recv = self._exchange(send)                                   # [r][c][k1 local]  ==  [j2][k1 local]
            rows = x.view(-1)[:k * n2 * 4].view(k, n2, 4) if x.numel() == k * n2 * 4 else torch.empty((k, n2, 4), dtype=x.dtype, device=x.device)
            self.local.io(2, recv, rows, k, False, NTT_TRANSPOSED, NTT_TRANSPOSED if natural_out else NTT_PLAIN)
            if natural_out and R > 1:                                     # rows holds [k2][k1 local]: block s = the k2 of rank s
                got = self._exchange(rows.view(R, c, k, 4))               # [r][k2 local][k1 local of r]
                return got.permute(1, 0, 2, 3).contiguous().view(c * n1, 4)   # [k2 local][k1]: index k1 + n1 k2
            return rows.view(-1, 4) if natural_out else rows
        self.local.ntt_rows(x, 1, False)                              # [c][j1] -> [c][k1]
        self.local.twiddle(x, self.rank * c, False)                   # * w_n^(j2 k1), j2 = rank*c + c_local
        send = x.view(c, R, k, 4).permute(1, 0, 2, 3).contiguous()    # [s][c][k1 local of s]
        recv = self._exchange(send)                                   # [r][c][k1 local]  ==  [j2][k1 local]
        rows = recv.view(n2, k, 4).permute(1, 0, 2).contiguous()      # [k1 local][j2]
        self.local.ntt_rows(rows, 2, False)                           # [k1 local][k2]
        return self.bc_out_to_natural(rows) if natural_out else rows

    def inverse(self, y, natural_in=False, natural_out=False):
        """y: (k, n2, 4), a BC(n1) block (used as scratch) -> (c, n1, 4), the BC(n2) block of the inverse transform
        (1/n included).  natural_in / natural_out: as in forward (contiguous slices of X in, of x out)."""
        R, n1, n2, c, k = self.world, self.n1, self.n2, self.c, self.k
        fused = getattr(self.local, "fused", False)
        if (natural_in or natural_out) and R == 1 and not fused:
            raise ValueError("DistNtt: on one rank natural-order input / output is offered with the GPU kernels only")
        if natural_in and R > 1:
            send = y.reshape(c, R, k, 4).permute(1, 0, 2, 3).contiguous()   # [s][k2 local][k1 local of s]
            y = self._exchange(send)                                        # [r][k2 local of r][k1 local]  ==  [k2][k1 local]: transposed BC(n1)
            if not fused:
                y = y.view(n2, k, 4).permute(1, 0, 2).contiguous()          # [k1 local][k2]
        if fused:
            import torch
            from ._lib import NTT_BLOCKED_TW, NTT_PLAIN, NTT_TRANSPOSED
            send = torch.empty((R, c, k, 4), dtype=y.dtype, device=y.device)     # [s][c local of s][k1 local]
            self.local.io(2, y, send, k, True, NTT_TRANSPOSED if natural_in else NTT_PLAIN, NTT_TRANSPOSED)   # (1/n2 applied)
            recv = self._exchange(send)                                   # [r][c][k1 local of r]: the blocked layout of [c][k1]
            cols = y.view(-1)[:c * n1 * 4].view(c, n1, 4)
            self.local.io(1, recv, cols, c, True, NTT_BLOCKED_TW, NTT_TRANSPOSED if natural_out else NTT_PLAIN, k.bit_length() - 1, self.rank * c)
            if natural_out and R > 1:                                     # cols holds [j1][j2 local]: block s = the j1 of rank s
                got = self._exchange(cols.view(R, k, c, 4))               # [r][j1 local][j2 local of r]
                return got.permute(1, 0, 2, 3).contiguous().view(k * n2, 4)   # [j1 local][j2]: index j1 n2 + j2
            return cols.view(-1, 4) if natural_out else cols              # * w_n^(-j2 k1), then [c][j1]   (1/n1 applied)
        self.local.ntt_rows(y, 2, True)                               # [k1 local][k2] -> [k1 local][j2]   (1/n2 applied)
        send = y.view(k, R, c, 4).permute(1, 2, 0, 3).contiguous()    # [s][c local of s][k1 local]
        recv = self._exchange(send)                                   # [r][c][k1 local of r]
        cols = recv.permute(1, 0, 2, 3).contiguous().view(c, n1, 4)   # [c][k1]
        self.local.twiddle(cols, self.rank * c, True)                 # * w_n^(-j2 k1)
        self.local.ntt_rows(cols, 1, True)                            # [c][j1]          (1/n1 applied)
        return self.bc_in_to_natural(cols) if natural_out else cols
