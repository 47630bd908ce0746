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
        back = d.inverse(y)
        ok_i = np.array_equal(back.cpu().numpy().view(np.uint64), d.scatter_in(full))
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
