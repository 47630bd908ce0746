"""GPU test (-m gpu): Groth16 prove at scale (roots-of-unity QAP, H by NTT, queries by MSM; SURVEY.md
section 8 row A7 / BASELINE.json configs[3]) against the closed-form scalars the known toxic waste
gives (zkp/groth16/test.py:303-325), and H(x) against the oracle's schoolbook multiply + long division
(zkp/groth16/poly_utils.py:17-45) at a size the oracle can reach."""
import numpy as np
import pytest

import c_oracle as co
import py_ref as o
from helpers import r1cs_closed_form, r1cs_crs_scalars
from zkhip import _lib
from zkhip.groth16.prover_ntt import BoolChainCircuit, ChainCircuit, ScaleCRS, ScaleProver

pytestmark = pytest.mark.gpu


def _dev(ints):
    import torch
    return torch.from_numpy(_lib.ints_to_limbs(ints).view(np.int64)).cuda()


TOXIC = dict(alpha=3926, beta=3604, gamma=2971, delta=1357)


def _ints(pt):
    """A proof point as plain integers: G1 -> (x, y), G2 -> ((x0, x1), (y0, y1)) -- the oracle's point format."""
    if pt is None:
        return None
    if hasattr(pt[0], "coeffs"):
        return tuple(tuple(int(c) for c in v.coeffs) for v in pt)
    return (int(pt[0]), int(pt[1]))


def _oracle_proof(circ, x_val, w, r, s):
    """(A*G1, B*G2, C*G1) with the scalars AND the points from the oracle (zkp/groth16/test.py:303-325); nothing of the
    library's CRS, field layer or group kernels is involved in the expectation."""
    A, B, C = r1cs_closed_form(circ.r1cs_csr(), w, circ.pub, dict(TOXIC, x=x_val), r, s)
    return co.g1_mul(o.G1, A), co.g2_mul(o.G2, B), co.g1_mul(o.G1, C)


def _check_crs_against_oracle(crs, x_val):
    """A few elements of every device-built query (zkp/groth16/setup.py:18-69) against the oracle's k*G."""
    circ = crs.circuit
    m, W = circ.m, circ.num_wires
    if m <= 1024:                                                        # every element of every query
        i12, i14, i15 = list(range(m)), [i for i in range(W) if i not in circ.pub], list(range(m - 1))
    else:
        i12, i14, i15 = [0, 1, m // 3, m - 1], [2, 3, W // 2, W // 2 + 1, W - 2, W - 1], [0, 1, m // 2, m - 2]
    k12, k14, k15 = r1cs_crs_scalars(circ.r1cs_csr(), dict(TOXIC, x=x_val), i12, i14, i15)
    rows = lambda t, idx: t[idx].cpu().numpy().view(np.uint64)
    assert np.array_equal(rows(crs.d_s12, i12), co.g1_fixed_base_arr(o.G1, co.to_limbs(k12)))
    assert np.array_equal(rows(crs.d_s22, i12), co.g2_fixed_base_arr(o.G2, co.to_limbs(k12)))
    assert np.array_equal(rows(crs.d_s14, i14), co.g1_fixed_base_arr(o.G1, co.to_limbs(k14)))
    assert np.array_equal(rows(crs.d_s15, i15), co.g1_fixed_base_arr(o.G1, co.to_limbs(k15)))
    assert not rows(crs.d_s14, circ.pub).any()                          # placeholders at the public wires (setup.py:50)
    t = TOXIC
    consts = [t["alpha"], t["delta"], t["beta"]]                         # the constant-term bases appended to sigma1_2 / sigma2_2
    assert np.array_equal(rows(crs.d_s12, [m, m + 1, m + 2]), co.g1_fixed_base_arr(o.G1, co.to_limbs(consts)))
    assert np.array_equal(rows(crs.d_s22, [m, m + 1]), co.g2_fixed_base_arr(o.G2, co.to_limbs([t["beta"], t["delta"]])))


@pytest.mark.parametrize("kind,log_m", [("chain", 4), ("chain", 10), ("chain", 13), ("bool", 1), ("bool", 5), ("bool", 10), ("bool", 13)])
def test_scale_prover_closed_form(kind, log_m):
    import torch
    circ = (ChainCircuit if kind == "chain" else BoolChainCircuit)(log_m, seed=3)
    x_val = 3721 + (1 << 200)
    crs = ScaleCRS(circ, x_val=x_val, **TOXIC)
    _check_crs_against_oracle(crs, x_val)
    w, a, b, c = circ.witness()
    assert all(a[k] * b[k] % o.R == c[k] for k in range(0, circ.m, max(1, circ.m // 64)))  # R1CS satisfied
    prover = ScaleProver(crs)
    r, s = 4106, 4565
    d_a, d_b, d_c, d_w = _dev(a), _dev(b), _dev(c), _dev(w)
    pa, pb, pc, h = prover.prove(d_a, d_b, d_c, d_w, r, s)
    assert (_ints(pa), _ints(pb), _ints(pc)) == _oracle_proof(circ, x_val, w, r, s)
    # the same proof from the witness alone: A.w, B.w, C.w by the device mat-vec (prove() consumed d_a..d_c in place)
    prover.load_r1cs(circ.r1cs_csr())
    qa, qb, qc, _ = prover.prove_from_witness(d_w, r, s)
    assert (qa, qb, qc) == (pa, pb, pc)
    torch.cuda.synchronize()
    hc = _lib.limbs_to_ints(h.cpu().numpy().view(np.uint64))
    assert hc[circ.m - 1] == 0                                   # deg H <= m - 2
    if log_m <= 5:
        # H against the reference's algorithm: (u_A * u_B - u_C) div (x^m - 1), remainder 0
        m = circ.m
        w_m = o.get_root_of_unity(m)
        uA, uB, uC = o.ifft(a, w_m), o.ifft(b, w_m), o.ifft(c, w_m)
        P = o.subtract_polys(o.multiply_polys(uA, uB), uC)
        Z = [(-1) % o.R] + [0] * (m - 1) + [1]
        q, rem = o.div_polys(P, Z)
        assert all(v == 0 for v in rem)
        assert hc[:len(q)] == q


def test_setup_with_many_levels_of_run_splitting(monkeypatch):
    """A wire that sits in every constraint (`one` in the chain circuit's C matrix) is summed level by level in runs of at most
    SPLIT entries per thread (prover_ntt._transposed_times); with SPLIT = 4 a 2^10-constraint circuit takes five levels, what a
    2^22-constraint one takes two of at the real SPLIT.  Every element of every query against the oracle."""
    from zkhip.groth16 import prover_ntt
    monkeypatch.setattr(prover_ntt, "SPLIT", 4)
    for circ in (ChainCircuit(10, seed=9), BoolChainCircuit(10, seed=9)):
        x_val = 3721 + (1 << 199)
        _check_crs_against_oracle(ScaleCRS(circ, x_val=x_val, **TOXIC), x_val)


def test_scale_prover_on_a_side_stream():
    """prove(..., stream=s) with s different from torch's current stream: the torch copies inside the prover and the backend's
    kernels must be ordered on s (they used to run on two streams).  The inputs are produced on s right before the call, so
    a prover that read them from the default stream's point of view would see stale data."""
    import torch
    circ = ChainCircuit(12, seed=5)
    x_val = 3721 + (1 << 200)
    crs = ScaleCRS(circ, x_val=x_val, **TOXIC)
    w, a, b, c = circ.witness()
    prover = ScaleProver(crs)
    prover.load_r1cs(circ.r1cs_csr())
    want = _oracle_proof(circ, x_val, w, 4106, 4565)
    side = torch.cuda.Stream()
    staging = _dev(w)
    d_w = torch.zeros_like(staging)
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(side):
            d_w.zero_()
            d_w.copy_(staging)                                  # the witness appears on the side stream only
        got = prover.prove_from_witness(d_w, 4106, 4565, stream=side.cuda_stream)
        assert tuple(_ints(p) for p in got[:3]) == want
    torch.cuda.synchronize()


def test_scale_prover_2pow20_constraints_closed_form():
    """BASELINE.json configs[3] at its full size: Groth16 prove() on a synthetic 2^20-constraint R1CS, witness resident in HBM
    -> (A, B, C), against the closed-form scalars of the known toxic waste; the CRS queries are bound to the MSM plans as
    bench.py does, and a second proof from the same prover must be identical (no state leaks between proofs)."""
    import torch
    log_m = 20
    circ = ChainCircuit(log_m, seed=7)
    w, a, b, c = circ.witness()
    x_val = 3721 + (1 << 201)
    crs = ScaleCRS(circ, x_val=x_val, **TOXIC)
    _check_crs_against_oracle(crs, x_val)                                # the device-built CRS, a few elements of every query
    prover = ScaleProver(crs)
    prover.load_r1cs(circ.r1cs_csr())
    d_w = _dev(w)
    r, s = 4106, 4565
    pa, pb, pc, h = prover.prove_from_witness(d_w, r, s)
    # expected points: scalars by the oracle's inverse NTT + Horner, points by the oracle's double-and-add
    assert (_ints(pa), _ints(pb), _ints(pc)) == _oracle_proof(circ, x_val, w, r, s)
    torch.cuda.synchronize()
    assert torch.equal(prover.abc[0], _dev(a)) and torch.equal(prover.abc[1], _dev(b))   # A.w and B.w of the device mat-vec
    assert (pa, pb, pc) == prover.prove_from_witness(d_w, r, s)[:3]
    # From 2^21 constraints on the merged proof_C query exceeds one 2^22-point chunk and is submitted while the A query is still
    # in flight on the same plan: the same situation here with 2^21-point chunks (the A query, m + 3 bases, still fits one; the
    # merged query's 3 m + 4 bases are a chunk and a half).
    del prover
    torch.cuda.empty_cache()
    chunked = ScaleProver(crs, chunk_log=21)
    chunked.load_r1cs(circ.r1cs_csr())
    assert (pa, pb, pc) == chunked.prove_from_witness(d_w, r, s)[:3]
    # From 2^22 constraints on the A query itself (m + 3 bases) is chunked as well, and a plan holds one chunked submission at a
    # time: the same with 2^20-point chunks (round 4: a 2^22-constraint proof failed with "a chunked submission is outstanding").
    del chunked
    torch.cuda.empty_cache()
    chunked = ScaleProver(crs, chunk_log=20)
    chunked.load_r1cs(circ.r1cs_csr())
    assert chunked.a_chunked and (pa, pb, pc) == chunked.prove_from_witness(d_w, r, s)[:3]


def _sharded_worker(rank, world, port, log_m, ret):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "..", "interactive-zkp-study_amd"))
    sys.path.insert(0, os.path.join(here, "..", "oracle"))
    import torch
    import torch.distributed as dist
    from zkhip.groth16.prover_ntt import ChainCircuit, ScaleCRS, ScaleProver, ShardedScaleProver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        circ = ChainCircuit(log_m, seed=3)
        crs = ScaleCRS(circ, alpha=3926, beta=3604, gamma=2971, delta=1357, x_val=3721 + (1 << 200))
        w, a, b, c = circ.witness()
        r, s = 4106, 4565
        single = ScaleProver(crs).prove(_dev(a), _dev(b), _dev(c), _dev(w), r, s)[:3]
        sharded = ShardedScaleProver(crs).prove(_dev(a), _dev(b), _dev(c), _dev(w), r, s)[:3]
        ret[rank] = bool(single == sharded)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_prover_equals_single_gpu(world):
    """Every MSM of the proof sharded over `world` ranks (rehearsed on one GPU over gloo): the proof is identical."""
    import os
    import torch.multiprocessing as mp
    port = 33500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_worker, args=(world, port, 10, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}
