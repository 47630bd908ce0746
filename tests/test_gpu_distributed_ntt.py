"""GPU tests of the multi-GPU single large NTT (zkhip.distributed.DistNtt): the batched
transform and the 2-D twiddle kernel through the C ABI, the four-step path on one rank against the direct plan
(bit-exact), and 2 / 4 ranks rehearsed on ONE GPU (gloo, host-staged exchange) against the direct plan."""
import os
import sys

import numpy as np
import pytest

import c_oracle as co
import py_ref as o
from helpers import rand_fr_limbs
from zkhip.device import NttPlan

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _direct(full, log_n, inverse=False):
    import torch
    d = torch.from_numpy(full.view(np.int64).copy()).cuda()
    NttPlan(log_n).run(d.data_ptr(), inverse, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("log_n,batch", [(0, 3), (3, 5), (8, 7), (10, 4), (11, 33), (12, 512), (16, 3)])
def test_batched_ntt_matches_single(log_n, batch):
    import torch
    rng = np.random.default_rng(50 + log_n)
    n = 1 << log_n
    X = rand_fr_limbs(rng, n * batch).reshape(batch, n, 4)
    st = torch.cuda.current_stream().cuda_stream
    plan = NttPlan(log_n)
    for inverse in (False, True):
        d = torch.from_numpy(X.view(np.int64).copy()).cuda()
        plan.run_batch(d.data_ptr(), batch, inverse, st)
        got = d.cpu().numpy().view(np.uint64)
        for b in sorted({0, batch // 2, batch - 1}):
            assert np.array_equal(got[b], _direct(X[b], log_n, inverse)), (b, inverse)
        if log_n <= 10:
            omega = pow(5, (o.R - 1) >> log_n, o.R)
            assert np.array_equal(got[batch - 1], co.ntt_arr(X[batch - 1], omega, inverse))


def test_twiddle_2d_kernel():
    import torch
    rng = np.random.default_rng(61)
    log_n, log_cols, rows, row0 = 14, 6, 9, 37
    X = rand_fr_limbs(rng, rows << log_cols)
    plan = NttPlan(log_n)
    w = pow(5, (o.R - 1) >> log_n, o.R)
    for inverse in (False, True):
        d = torch.from_numpy(X.view(np.int64).copy()).cuda()
        plan.twiddle(d.data_ptr(), log_cols, rows, row0, inverse, torch.cuda.current_stream().cuda_stream)
        got = co.from_limbs(d.cpu().numpy().view(np.uint64))
        ww = pow(w, -1, o.R) if inverse else w
        vals = co.from_limbs(X)
        want = [vals[b * 64 + k] * pow(ww, (row0 + b) * k, o.R) % o.R for b in range(rows) for k in range(64)]
        assert got == want


@pytest.mark.parametrize("log_n,batch,log_block,big_log", [(1, 2, 0, 2), (4, 8, 2, 9), (6, 16, 3, 12), (8, 12, 8, 13), (10, 64, 7, 16), (11, 32, 9, 16), (12, 256, 10, 20),
                                                           (13, 8, 5, 17), (16, 4, 14, 18), (17, 4, 10, 19), (19, 2, 3, 20)])   # the last two: three-pass transforms
def test_ntt_io_layouts_against_the_separate_passes(log_n, batch, log_block, big_log):
    """zk_ntt_dev_io: the four-step layouts read by the first loads / written by the last stores must give exactly what the
    separate passes give -- batched transform (zk_ntt_dev_batch), 2-D twiddle (zk_ntt_twiddle_dev) and a numpy permutation."""
    import torch
    from zkhip._lib import NTT_BLOCKED_TW, NTT_PLAIN, NTT_TRANSPOSED
    rng = np.random.default_rng(900 + log_n)
    n, kb = 1 << log_n, 1 << log_block
    row0 = ((1 << big_log) >> log_n) - batch                               # the last rows of the large transform
    assert row0 >= 0
    X = rand_fr_limbs(rng, n * batch).reshape(batch, n, 4)
    st = torch.cuda.current_stream().cuda_stream
    plan, big = NttPlan(log_n), NttPlan(big_log)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    host = lambda t: t.cpu().numpy().view(np.uint64)
    blocked = lambda a: np.ascontiguousarray(a.reshape(batch, n // kb, kb, 4).transpose(1, 0, 2, 3)).reshape(batch, n, 4)
    unblocked = lambda a: np.ascontiguousarray(a.reshape(n // kb, batch, kb, 4).transpose(1, 0, 2, 3)).reshape(batch, n, 4)
    transposed = lambda a: np.ascontiguousarray(a.transpose(1, 0, 2))       # (batch, n, 4) -> (n, batch, 4)
    for inverse in (False, True):
        rows = dev(X)
        plan.run_batch(rows.data_ptr(), batch, inverse, st)
        want_rows = host(rows).copy()                                       # the plain batched transform
        tw = dev(want_rows)
        big.twiddle(tw.data_ptr(), log_n, batch, row0, inverse, st)
        want_tw = host(tw).copy()                                           # ... times w_N^(+-(row0 + b) i)
        out = torch.zeros_like(rows)
        # plain in, blocked + twiddle out (forward step 1 of the four-step transform)
        plan.run_io(dev(X).data_ptr(), out.data_ptr(), batch, inverse, NTT_PLAIN, NTT_BLOCKED_TW, log_block, row0, big, inverse, st)
        assert np.array_equal(host(out).reshape(batch, n, 4), blocked(want_tw)), ("out blocked", inverse)
        # transposed in (forward step 3), plain / transposed out
        plan.run_io(dev(transposed(X)).data_ptr(), out.data_ptr(), batch, inverse, NTT_TRANSPOSED, NTT_PLAIN, 0, 0, None, False, st)
        assert np.array_equal(host(out).reshape(batch, n, 4), want_rows), ("in transposed", inverse)
        plan.run_io(dev(transposed(X)).data_ptr(), out.data_ptr(), batch, inverse, NTT_TRANSPOSED, NTT_TRANSPOSED, 0, 0, None, False, st)
        assert np.array_equal(host(out).reshape(n, batch, 4), transposed(want_rows)), ("in / out transposed", inverse)
        plan.run_io(dev(X).data_ptr(), out.data_ptr(), batch, inverse, NTT_PLAIN, NTT_TRANSPOSED, 0, 0, None, False, st)
        assert np.array_equal(host(out).reshape(n, batch, 4), transposed(want_rows)), ("out transposed", inverse)
        # blocked + twiddle in (inverse step 3): the twiddle comes BEFORE the transform
        pre = dev(X)
        big.twiddle(pre.data_ptr(), log_n, batch, row0, inverse, st)
        plan.run_batch(pre.data_ptr(), batch, inverse, st)
        plan.run_io(dev(blocked(X)).data_ptr(), out.data_ptr(), batch, inverse, NTT_BLOCKED_TW, NTT_PLAIN, log_block, row0, big, inverse, st)
        assert np.array_equal(host(out).reshape(batch, n, 4), host(pre)), ("in blocked", inverse)
        plan.run_io(dev(blocked(X)).data_ptr(), out.data_ptr(), batch, inverse, NTT_BLOCKED_TW, NTT_TRANSPOSED, log_block, row0, big, inverse, st)
        assert np.array_equal(host(out).reshape(n, batch, 4), transposed(host(pre))), ("in blocked, out transposed", inverse)
        plan.run_io(dev(transposed(X)).data_ptr(), out.data_ptr(), batch, inverse, NTT_TRANSPOSED, NTT_BLOCKED_TW, log_block, row0, big, inverse, st)
        assert np.array_equal(host(out).reshape(batch, n, 4), blocked(want_tw)), ("in transposed, out blocked", inverse)
    d = dev(X)
    with pytest.raises(Exception):
        plan.run_io(d.data_ptr(), d.data_ptr(), batch, False, NTT_PLAIN, NTT_TRANSPOSED, 0, 0, None, False, st)    # layout change in place
    with pytest.raises(Exception):
        plan.run_io(d.data_ptr(), out.data_ptr(), batch, False, NTT_PLAIN, NTT_BLOCKED_TW, log_block, 0, None, False, st)   # no large plan
    assert unblocked(blocked(X)).tobytes() == X.tobytes()


@pytest.mark.parametrize("log_n,l1", [(2, None), (9, 3), (12, None), (13, 6), (16, None), (20, None), (21, 11)])
def test_four_step_single_rank_matches_direct(log_n, l1):
    """world 1: the whole vector in residue-major storage; forward() must equal the direct plan bit for bit."""
    import torch
    from zkhip.distributed import DistNtt
    rng = np.random.default_rng(70 + log_n)
    n = 1 << log_n
    full = rand_fr_limbs(rng, n)
    d = DistNtt(log_n, l1=l1)
    x = torch.from_numpy(d.scatter_in(full).view(np.int64)).cuda()
    y = d.forward(x)
    assert np.array_equal(y.cpu().numpy().view(np.uint64), d.scatter_out(_direct(full, log_n)))
    back = d.inverse(y)
    assert np.array_equal(back.cpu().numpy().view(np.uint64), d.scatter_in(full))
    # natural order in and out (one rank): the four-step route as a whole against the direct plan, and the explicit
    # transform / twiddle / permute route (what the CPU tests run with the oracle as local transform) against both
    nat = torch.from_numpy(full.view(np.int64).copy()).cuda()
    fwd = d.forward(nat, natural_in=True, natural_out=True)
    assert np.array_equal(fwd.cpu().numpy().view(np.uint64).reshape(-1, 4), _direct(full, log_n))
    inv = d.inverse(fwd, natural_in=True, natural_out=True)
    assert np.array_equal(inv.cpu().numpy().view(np.uint64).reshape(-1, 4), full)

    class Unfused(type(d.local)):
        fused = False
    du = DistNtt(log_n, l1=l1, local=Unfused(log_n, d.l1, d.l2))
    yu = du.forward(torch.from_numpy(d.scatter_in(full).view(np.int64)).cuda())
    assert np.array_equal(yu.cpu().numpy().view(np.uint64), d.scatter_out(_direct(full, log_n)))
    assert np.array_equal(du.inverse(yu).cpu().numpy().view(np.uint64), d.scatter_in(full))


def _worker(rank, world, port, log_n, l1, ret):
    sys.path.insert(0, os.path.join(HERE, "..", "interactive-zkp-study_amd"))
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    from zkhip.distributed import DistNtt
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1 << log_n
        full = rand_fr_limbs(np.random.default_rng(80 + log_n), n)   # same vector on every rank
        want = _direct(full, log_n)
        d = DistNtt(log_n, l1=l1)
        x = torch.from_numpy(d.scatter_in(full).view(np.int64)).cuda()
        y = d.forward(x)
        ok_f = np.array_equal(y.cpu().numpy().view(np.uint64), d.scatter_out(want))
        per = n // world                                                  # natural order across the ranks: one more all-to-all
        nat = d.bc_out_to_natural(y.clone())
        ok_f = ok_f and np.array_equal(nat.cpu().numpy().view(np.uint64), want[rank * per:(rank + 1) * per])
        ok_f = ok_f and np.array_equal(d.natural_to_bc_out(nat).cpu().numpy().view(np.uint64), d.scatter_out(want))
        back = d.inverse(y)
        ok_i = np.array_equal(back.cpu().numpy().view(np.uint64), d.scatter_in(full))
        ok_i = ok_i and np.array_equal(d.bc_in_to_natural(back).cpu().numpy().view(np.uint64), full[rank * per:(rank + 1) * per])
        # natural order in and out with the transposes inside the pass kernels (ZK_NTT_TRANSPOSED stores = the send buffer)
        x_nat = torch.from_numpy(full[rank * per:(rank + 1) * per].copy().view(np.int64)).cuda()
        fwd_nat = d.forward(x_nat.clone(), natural_in=True, natural_out=True)
        ok_f = ok_f and np.array_equal(fwd_nat.cpu().numpy().view(np.uint64).reshape(-1, 4), want[rank * per:(rank + 1) * per])
        inv_nat = d.inverse(fwd_nat, natural_in=True, natural_out=True)
        ok_i = ok_i and np.array_equal(inv_nat.cpu().numpy().view(np.uint64).reshape(-1, 4), full[rank * per:(rank + 1) * per])
        mixed = d.forward(torch.from_numpy(d.scatter_in(full).view(np.int64)).cuda(), natural_out=True)
        ok_f = ok_f and np.array_equal(mixed.cpu().numpy().view(np.uint64).reshape(-1, 4), want[rank * per:(rank + 1) * per])
        mixed_i = d.inverse(torch.from_numpy(d.scatter_out(want).view(np.int64)).cuda(), natural_out=True)
        ok_i = ok_i and np.array_equal(mixed_i.cpu().numpy().view(np.uint64).reshape(-1, 4), full[rank * per:(rank + 1) * per])
        ret[rank] = (bool(ok_f), bool(ok_i))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,l1", [(2, 14, None), (4, 17, 8)])
def test_dist_ntt_ranks_on_one_gpu(world, log_n, l1):
    import torch.multiprocessing as mp
    port = 32500 + (os.getpid() % 2000) + log_n
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, log_n, l1, ret), nprocs=world, join=True)
    assert dict(ret) == {r: (True, True) for r in range(world)}
