"""World-size-2/4 `gloo` tests (CPU) of the multi-GPU single large NTT (zkhip.distributed.DistNtt): the block-cyclic
layouts, the four-step index algebra and the one all-to-all are the product code under test; the local transforms and
the twiddle come from the oracle here (test infrastructure), the HIP kernels take their place on a GPU
(tests/test_gpu_distributed_ntt.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


class OracleLocal:
    """ntt_rows / twiddle of DistNtt on CPU tensors, computed by the C oracle."""

    def __init__(self, co, o, log_n):
        self.co, self.o, self.log_n = co, o, log_n

    def _omega(self, log_len, inverse):
        w = pow(5, (self.o.R - 1) >> log_len, self.o.R)
        return w

    def ntt_rows(self, t, which, inverse):
        a = t.numpy().view(np.uint64)
        for b in range(a.shape[0]):
            n = a.shape[1]
            a[b] = self.co.ntt_arr(a[b].copy(), self._omega(n.bit_length() - 1, inverse), inverse)

    def twiddle(self, t, row0, inverse):
        R = self.o.R
        w = pow(5, (R - 1) >> self.log_n, R)
        if inverse:
            w = pow(w, -1, R)
        a = t.numpy().view(np.uint64)
        for b in range(a.shape[0]):
            vals = self.co.from_limbs(a[b])
            a[b] = self.co.to_limbs([v * pow(w, (row0 + b) * k, R) % R for k, v in enumerate(vals)])


def _worker(rank, world, port, log_n, l1, ret):
    sys.path.insert(0, os.path.join(HERE, "..", "interactive-zkp-study_amd"))
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    import c_oracle as co
    import py_ref as o
    from zkhip.distributed import DistNtt

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1 << log_n
        rng = np.random.default_rng(7)  # same vector on every rank; each rank keeps its own block
        full = co.to_limbs([int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(n)])
        omega = pow(5, (o.R - 1) >> log_n, o.R)
        want = co.ntt_arr(full.copy(), omega, False)
        d = DistNtt(log_n, l1=l1, local=OracleLocal(co, o, log_n))
        x = torch.from_numpy(d.scatter_in(full).view(np.int64))
        assert tuple(x.shape[:2]) == d.local_shape_in()
        y = d.forward(x)
        ok_f = np.array_equal(y.numpy().view(np.uint64), d.scatter_out(want)) and tuple(y.shape[:2]) == d.local_shape_out()
        # natural order across the ranks: one more all-to-all each way
        per = n // world
        nat = d.bc_out_to_natural(y.clone())
        ok_n = np.array_equal(nat.numpy().view(np.uint64), want[rank * per:(rank + 1) * per])
        ok_n = ok_n and np.array_equal(d.natural_to_bc_out(nat).numpy().view(np.uint64), d.scatter_out(want))
        x_nat = torch.from_numpy(full[rank * per:(rank + 1) * per].copy().view(np.int64))
        ok_n = ok_n and np.array_equal(d.natural_to_bc_in(x_nat).numpy().view(np.uint64), d.scatter_in(full))
        back = d.inverse(y)
        ok_i = np.array_equal(back.numpy().view(np.uint64), d.scatter_in(full))
        ok_i = ok_i and np.array_equal(d.bc_in_to_natural(back).numpy().view(np.uint64), full[rank * per:(rank + 1) * per])
        # the transforms themselves from and to the natural order across the ranks (contiguous slices in, contiguous slices out)
        fwd_nat = d.forward(x_nat.clone(), natural_in=True, natural_out=True)
        ok_n = ok_n and np.array_equal(fwd_nat.numpy().view(np.uint64).reshape(-1, 4), want[rank * per:(rank + 1) * per])
        inv_nat = d.inverse(fwd_nat, natural_in=True, natural_out=True)
        ok_i = ok_i and np.array_equal(inv_nat.numpy().view(np.uint64).reshape(-1, 4), full[rank * per:(rank + 1) * per])
        mixed = d.forward(torch.from_numpy(d.scatter_in(full).view(np.int64)), natural_out=True)       # block-cyclic in, natural out
        ok_n = ok_n and np.array_equal(mixed.numpy().view(np.uint64).reshape(-1, 4), want[rank * per:(rank + 1) * per])
        # several vectors with their exchanges in flight together: the same results as one by one
        fulls = [co.to_limbs([int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(n)]) for _ in range(3)]
        wants = [co.ntt_arr(f.copy(), omega, False) for f in fulls]
        ys = d.forward_many([torch.from_numpy(d.scatter_in(f).view(np.int64)) for f in fulls])
        ok_f = ok_f and all(np.array_equal(yy.numpy().view(np.uint64), d.scatter_out(w)) for yy, w in zip(ys, wants))
        xs = d.inverse_many([yy.clone() for yy in ys])
        ok_i = ok_i and all(np.array_equal(xx.numpy().view(np.uint64), d.scatter_in(f)) for xx, f in zip(xs, fulls))
        ret[rank] = (bool(ok_f and ok_n), bool(ok_i))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,l1", [(2, 6, None), (2, 7, 3), (4, 8, None), (4, 7, 4), (8, 8, None)])   # 8: the driver's node size
def test_dist_ntt_gloo(world, log_n, l1):
    port = 31500 + (os.getpid() % 2000) + 7 * log_n + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, log_n, l1, ret), nprocs=world, join=True)
    assert dict(ret) == {r: (True, True) for r in range(world)}


class NullLocal:
    """Local transforms that do nothing: the dry run below is about the shapes and byte counts of the exchanges."""

    def ntt_rows(self, t, which, inverse):
        pass

    def twiddle(self, t, row0, inverse):
        pass


def _dry_worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(HERE, "..", "interactive-zkp-study_amd"))
    from zkhip.distributed import DistNtt, all_gather_partials
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sent = []
        real = dist.all_to_all_single

        def counting(out, inp, *a, **kw):
            sent.append(inp.numel() * inp.element_size())
            return real(out, inp, *a, **kw)
        dist.all_to_all_single = counting
        # (1) ONE transform of 2^24 points over the eight ranks: forward, inverse, and from / to the natural order
        d = DistNtt(24, local=NullLocal())
        n_loc = (1 << 24) // world
        assert d.c * d.n1 == n_loc == d.k * d.n2
        x = torch.zeros(d.local_shape_in() + (4,), dtype=torch.int64)
        y = d.forward(x)
        assert tuple(y.shape) == d.local_shape_out() + (4,)
        back = d.inverse(y)
        assert tuple(back.shape) == d.local_shape_in() + (4,)
        nat = d.forward(torch.zeros((n_loc, 4), dtype=torch.int64), natural_in=True, natural_out=True)
        assert tuple(nat.shape) == (n_loc, 4)
        assert tuple(d.inverse(nat, natural_in=True, natural_out=True).shape) == (n_loc, 4)
        per_exchange = n_loc * 32                         # every rank sends its whole block, 1/R of it to each peer
        assert sent == [per_exchange] * (1 + 1 + 3 + 3), sent
        del x, y, back, nat
        # (2) the collectives of one DistScaleProver proof at 2^20 constraints (groth16/prover_dist.py prove): three inverse
        #     transforms, three forward ones on the coset, one inverse -- one all-to-all each -- then ONE all-gather of the 64-limb partials
        sent.clear()
        dn = DistNtt(20, local=NullLocal())
        cn = dn.c * dn.n1
        assert cn == dn.k * dn.n2 == (1 << 20) // world
        shape_ev, shape_co = (dn.k, dn.n2, 4), (dn.c, dn.n1, 4)
        coef = [u.reshape(cn, 4) for u in dn.inverse_many([torch.zeros(shape_ev, dtype=torch.int64) for _ in range(3)])]
        on_coset = [v.reshape(cn, 4) for v in dn.forward_many([u.view(shape_co) for u in coef])]
        h = dn.inverse(on_coset[0].view(shape_ev)).reshape(cn, 4)
        assert tuple(h.shape) == (cn, 4) and sent == [cn * 32] * 7, sent
        mine = np.full(64, rank + 1, dtype=np.uint64)     # 16 + 16 + 32 limbs: the partials of proof_A, proof_C, proof_B
        everyone = all_gather_partials(mine)
        assert everyone.shape == (world, 64) and [int(v) for v in everyone[:, 0]] == list(range(1, world + 1))
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_dry_run_of_every_collective_at_eight_ranks():
    """The driver's node has eight GPUs and no builder ever had more than one: every collective of the single 2^24-point transform
    (block-cyclic and natural order) and of a distributed Groth16 proof at 2^20 constraints, with the REAL shapes and byte counts,
    over gloo on eight CPU ranks and local transforms that do nothing -- a shape or count that does not fit fails here, not on the
    first eight-GPU run."""
    world = 8
    port = 33500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dry_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_dist_ntt_rejects_bad_world():
    sys.path.insert(0, os.path.join(HERE, "..", "interactive-zkp-study_amd"))
    from zkhip.distributed import DistNtt
    d = DistNtt(4, local=object())           # no process group: world 1
    assert (d.world, d.n1, d.n2, d.c, d.k) == (1, 4, 4, 4, 4) and d.local_shape_in() == (4, 4)
    with pytest.raises(ValueError):
        DistNtt(4, l1=5, local=object())
