"""GPU test (-m gpu): the Groth16 prover with every vector distributed over the ranks (zkhip.groth16.prover_dist; SURVEY.md
section 8 rows E1 + E2 as one chain: sparse mat-vec on the rank's rows -> distributed inverse / coset / forward transforms, ONE
all-to-all each -> pointwise quotient -> MSMs over the rank's coefficients against its slice of the queries -> ONE all-gather of
the partial sums).  The proof must equal the oracle's closed form (zkp/groth16/test.py:303-325, oracle/scale_ref.py) and the
single-GPU prover's, on one rank and with 2 and 4 ranks rehearsed on ONE GPU over gloo -- even and odd log m (the block-cyclic
layouts of evaluations and coefficients differ for odd log m), uniform and boolean-heavy witnesses."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOXIC = dict(alpha=3926, beta=3604, gamma=2971, delta=1357)
X_VAL = 3721 + (1 << 200)


def _ints(pt):
    if pt is None:
        return None
    if hasattr(pt[0], "coeffs"):
        return tuple(tuple(int(c) for c in v.coeffs) for v in pt)
    return (int(pt[0]), int(pt[1]))


def _prove_and_check(kind, log_m, compare_single):
    """Runs on every rank: distributed proof == oracle closed form (== the single-GPU prover's proof when asked)."""
    import torch
    import c_oracle as co
    import py_ref as o
    from scale_ref import r1cs_closed_form
    from zkhip import _lib
    from zkhip.groth16.prover_dist import DistScaleCRS, DistScaleProver
    from zkhip.groth16.prover_ntt import BoolChainCircuit, ChainCircuit, ScaleCRS, ScaleProver
    circ = (ChainCircuit if kind == "chain" else BoolChainCircuit)(log_m, seed=3)
    w = circ.witness()[0]
    d_w = torch.from_numpy(_lib.ints_to_limbs(w).view(np.int64)).cuda()
    r, s = 4106, 4565
    crs = DistScaleCRS(circ, x_val=X_VAL, **TOXIC)
    prover = DistScaleProver(crs)
    got = tuple(_ints(p) for p in prover.prove(d_w, r, s))
    again = tuple(_ints(p) for p in prover.prove(d_w, r, s))               # no state leaks between proofs
    A, B, C = r1cs_closed_form(circ.r1cs_csr(), w, circ.pub, dict(TOXIC, x=X_VAL), r, s)
    want = (co.g1_mul(o.G1, A), co.g2_mul(o.G2, B), co.g1_mul(o.G1, C))
    ok = got == want and again == want
    if compare_single:
        single = ScaleProver(ScaleCRS(circ, x_val=X_VAL, **TOXIC))
        single.load_r1cs(circ.r1cs_csr())
        ok = ok and tuple(_ints(p) for p in single.prove_from_witness(d_w, r, s)[:3]) == want
    # the rank holds 1/R of every query (plus the three constant bases and its wires of the L query), nothing more
    share = crs.n_g1 - 3 - (crs.w_hi - crs.w_lo)
    return bool(ok and share == 2 * circ.m // crs.world and crs.d_g2.shape[0] == circ.m // crs.world + 2)


@pytest.mark.parametrize("kind,log_m", [("chain", 2), ("chain", 10), ("bool", 11), ("chain", 14), ("bool", 18)])
def test_distributed_prover_on_one_rank(kind, log_m):
    assert _prove_and_check(kind, log_m, compare_single=log_m <= 14)


def _worker(rank, world, port, kind, log_m, ret):
    sys.path.insert(0, os.path.join(HERE, "..", "interactive-zkp-study_amd"))
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _prove_and_check(kind, log_m, compare_single=(rank == 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,log_m", [(2, "chain", 10), (2, "bool", 11), (4, "chain", 11), (4, "bool", 12)])
def test_distributed_prover_ranks_on_one_gpu(world, kind, log_m):
    import torch.multiprocessing as mp
    port = 34500 + (os.getpid() % 2000) + 8 * world + log_m
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, kind, log_m, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}
