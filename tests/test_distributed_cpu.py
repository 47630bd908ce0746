"""World-size-2 `gloo` test (CPU) of the multi-GPU MSM path used by bench.py --gpus N:
shard the points by contiguous chunks, all-gather the 128-byte partial sums, fold in rank order.
On CPU the per-rank partial comes from the oracle (test infrastructure); the sharding, the
collective and the host fold (zk_msm_fold_partials) are the product code under test."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, n, ret):
    sys.path.insert(0, os.path.join(HERE, "..", "interactive-zkp-study_amd"))
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    sys.path.insert(0, HERE)
    import c_oracle as co
    import py_ref as o
    from test_abi import xyzz_partial_g1
    from zkhip import _lib
    from zkhip.distributed import ExchangeWorker, shard_range, sharded_msm, sharded_msm_start

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(99)  # same data on every rank; each rank uses its own chunk
        sc = co.to_limbs([int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(n)])
        ks = co.to_limbs([int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(n)])
        pts = co.g1_fixed_base_arr(o.G1, ks)
        lo, hi = shard_range(n, rank, world)
        local = co.g1_from_arr(co.g1_msm_arr(sc[lo:hi], pts[lo:hi]))[0] if hi > lo else None
        got = sharded_msm(_lib.GROUP_G1, xyzz_partial_g1(local))
        full = co.g1_from_arr(co.g1_msm_arr(sc, pts))[0]
        ok = (got is None and full is None) or (got is not None and (int(got[0]), int(got[1])) == full)
        # the pipelined form bench.py uses: several exchanges started before the first is collected, collected in order
        handles = [sharded_msm_start(_lib.GROUP_G1, xyzz_partial_g1(local)) for _ in range(3)]
        ok = ok and all(h.done() for h in handles)            # the host path finishes inside the start call
        ok = ok and all(h.result() == got for h in handles)
        # and the thread that runs the exchange side of bench.py's multi-rank loop: posts in step order, flush folds the last one
        worker = ExchangeWorker(_lib.GROUP_G1)
        for step in range(7):
            worker.post(xyzz_partial_g1(local if step != 3 else None))   # step 3: this rank contributes infinity
        ok = ok and worker.flush() == got
        worker.post(xyzz_partial_g1(None))
        only_others = worker.flush()                                     # every rank posts infinity: the sum is infinity
        ok = ok and only_others is None
        try:
            worker.post("not a partial")                                 # fails before any collective, on every rank: surfaces at flush()
            worker.flush()
            ok = False
        except Exception:                                                # noqa: BLE001
            pass
        worker.post(xyzz_partial_g1(local))                              # and the worker is usable afterwards
        ok = ok and worker.flush() == got
        thread = worker._thread
        worker.post(xyzz_partial_g1(local))                              # close() finishes what is in flight, then ends the thread
        worker.close()
        ok = ok and worker.res == got and not thread.is_alive()
        worker.close()                                                   # idempotent
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1), (2, 37), (8, 37)])   # 8 ranks: BASELINE.json configs[4]'s rank count, on the CPU
def test_sharded_msm_gloo(world, n):
    port = 29500 + (os.getpid() % 2000) + n + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}
