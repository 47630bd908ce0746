"""GPU tests of the device-resident PLONK prover (zkhip.plonk.prover_device.DevicePlonk, SURVEY.md section 8 row f3 at
scale): with the same blinding scalars its proof and preprocessing commitments equal the list prover's (which mirrors
zkp/plonk/prover/round1..5.py) bit for bit; proofs of a 2^12-gate synthetic circuit verify and tampered ones do not."""
import copy

import numpy as np
import pytest

import py_ref as o
from zkhip import _lib
from zkhip.field import FR, g1_to_limbs
from zkhip.plonk.circuit import Circuit
from zkhip.plonk.permutation import build_permutation_polynomials
from zkhip.plonk.preprocessor import preprocess
from zkhip.plonk.prover import Proof, prove
from zkhip.plonk.prover_device import DevicePlonk
from zkhip.plonk.srs import SRS
from zkhip.plonk.verifier import verify

pytestmark = pytest.mark.gpu
R = o.R


def _limbs(vals):
    return _lib.ints_to_limbs([int(v) % R for v in vals])


def _chain_circuit(rows, seed):
    """Alternating multiplication / addition gates, each output wired to the next gate's left input."""
    rng = np.random.default_rng(seed)
    c = Circuit()
    a, b, cc = [], [], []
    cur = int(rng.integers(2, 1 << 40))
    for i in range(rows):
        y = int(rng.integers(1, 1 << 40))
        if i % 2 == 0:
            g = c.add_multiplication_gate()
            out = cur * y % R
        else:
            g = c.add_addition_gate()
            out = (cur + y) % R
        a.append(FR(cur)); b.append(FR(y)); cc.append(FR(out))
        if i:
            c.add_copy_constraint(g - 1, 2, g, 0)
        cur = out
    return c, a, b, cc


def _device_from_circuit(circuit, pp, srs):
    sel = [_limbs(col) for col in circuit.get_selector_polynomials()]
    sig = [_limbs(col) for col in build_permutation_polynomials(pp.sigma, pp.n, pp.domain)]
    return DevicePlonk(sel, sig, g1_to_limbs(srs.g1_powers))


@pytest.mark.parametrize("which", ["toy", "chain13", "chain64"])
def test_device_prover_equals_list_prover(which):
    if which == "toy":
        circuit, a, b, c, pub = Circuit.x3_plus_x_plus_5_eq_35()     # n = 4: quotient domain 8n
    else:
        circuit, a, b, c = _chain_circuit(13 if which == "chain13" else 64, 5)
        pub = []
    srs = SRS.generate(len(circuit.gates) + 80, seed=42)
    pp = preprocess(circuit, srs)                                    # pads the circuit to a power of two
    n = pp.n
    pad = lambda col: list(col) + [FR(0)] * (n - len(col))
    a, b, c = pad(a), pad(b), pad(c)
    blinding = [1000 + 7 * i for i in range(9)]
    want = prove(circuit, a, b, c, pub, pp, srs, blinding=blinding)
    dev = _device_from_circuit(circuit, pp, srs)
    dpp = dev.preprocessed()
    for name in ("q_l", "q_r", "q_o", "q_m", "q_c", "s_sigma1", "s_sigma2", "s_sigma3"):
        assert getattr(dpp, name + "_comm") == getattr(pp, name + "_comm"), name
    got = dev.prove(_limbs(a), _limbs(b), _limbs(c), blinding=blinding)
    for f in Proof.FIELDS:
        assert getattr(got, f) == getattr(want, f), f
    assert verify(got, pub, dpp, srs)
    fresh = dev.prove(_limbs(a), _limbs(b), _limbs(c))                # random blinding: a different, valid proof
    assert fresh.a_comm != got.a_comm and verify(fresh, pub, dpp, srs)


def test_device_prover_2pow12_verifies_and_rejects_tampering():
    from zkhip.field import G1, G2, fixed_base_mul
    rng = np.random.default_rng(77)
    n = 1 << 12
    # synthetic circuit straight as arrays: gate i multiplies (even i) or adds (odd i); c_i is wired to a_{i+1}
    even = (np.arange(n) % 2 == 0)
    q_m = even.astype(object)
    q_l = (~even).astype(object)
    q_r = (~even).astype(object)
    q_o = np.array([R - 1] * n, dtype=object)
    q_c = np.array([0] * n, dtype=object)
    a, b, c = [0] * n, [0] * n, [0] * n
    cur = 3
    for i in range(n):
        y = int(rng.integers(1, 1 << 50))
        a[i], b[i] = cur, y
        c[i] = cur * y % R if i % 2 == 0 else (cur + y) % R
        cur = c[i]
    sigma = list(range(3 * n))
    for i in range(1, n):                                             # position = wire * n + gate
        p1, p2 = 2 * n + (i - 1), i
        sigma[p1], sigma[p2] = sigma[p2], sigma[p1]
    w = pow(5, (R - 1) // n, R)
    dom = [pow(w, i, R) for i in range(n)]
    label = lambda pos: dom[pos] if pos < n else (2 * dom[pos - n] % R if pos < 2 * n else 3 * dom[pos - 2 * n] % R)
    sig = [[label(sigma[col * n + i]) for i in range(n)] for col in range(3)]
    tau = 0xC0FFEE1234567
    powers, t = [], 1
    for _ in range(n + 8):
        powers.append(t)
        t = t * tau % R
    srs_limbs = g1_to_limbs(fixed_base_mul(G1, powers))

    class Srs:                                                         # what the verifier reads
        g2_powers = [G2] + fixed_base_mul(G2, [tau])
    dev = DevicePlonk([_limbs(q) for q in (q_l, q_r, q_o, q_m, q_c)], [_limbs(s) for s in sig], srs_limbs)
    proof = dev.prove(_limbs(a), _limbs(b), _limbs(c))
    pp = dev.preprocessed()
    assert verify(proof, [], pp, Srs)
    # independent of this repo's verifier: tau is known, so every commitment must be p(tau) * G1 -- Horner on Python integers over
    # the downloaded coefficients, and the device's own scale-and-sum evaluation, against one scalar multiplication each
    # the expected point is the C ORACLE's scalar multiplication, so nothing of it comes from libzkhip
    import c_oracle as co
    oracle_mul = lambda k: co.g1_mul(o.G1, int(k) % o.R)
    assert dev.closed_form_mismatches(proof, tau, on_host=True, mul=oracle_mul) == []
    assert dev.closed_form_mismatches(proof, tau, mul=oracle_mul) == []
    assert dev.closed_form_mismatches(proof, tau) == []               # the library's own group operation agrees
    wrong = copy.copy(proof)
    wrong.t_hi_comm = proof.t_mid_comm
    assert dev.closed_form_mismatches(wrong, tau, mul=oracle_mul) == ["t_hi_comm"]
    bad = copy.copy(proof)
    bad.a_eval = proof.a_eval + FR(1)
    assert not verify(bad, [], pp, Srs)
    bad = copy.copy(proof)
    bad.t_mid_comm = proof.t_lo_comm
    assert not verify(bad, [], pp, Srs)
    c_bad = list(c)
    c_bad[100] = (c_bad[100] + 1) % R                                   # unsatisfied gate: t is not a polynomial
    with pytest.raises(ValueError):
        dev.prove(_limbs(a), _limbs(b), _limbs(c_bad))


def test_fused_permutation_factors_equal_the_vector_form_and_the_oracle():
    """zk_plonk_perm_factors_dev (one pass over a b c | s1 s2 s3 | x) against the ten lincomb / product launches it replaced and
    against Python integers (permutation.py:118-131), n = 64 and an odd length, beta / gamma near the field's edges."""
    import torch
    from zkhip.device import plonk_perm_factors
    circuit, a, b, c = _chain_circuit(64, 11)
    srs = SRS.generate(64 + 8, seed=42)
    pp = preprocess(circuit, srs)
    n = pp.n
    dev = _device_from_circuit(circuit, pp, srs)
    cols = [torch.from_numpy(_limbs(col).view(np.int64)).cuda() for col in (a, b, c)]
    s1, s2, s3 = build_permutation_polynomials(pp.sigma, n, pp.domain)
    for be, ga in ((0x1234567, 0x89ABCDEF), (R - 1, R - 2), (0, 5), (7, 0)):
        num, den = dev._accumulator_factors(cols, be, ga)
        got = [_lib.limbs_to_ints(t.cpu().numpy().view(np.uint64)) for t in (num, den)]
        un, ud = dev._accumulator_factors_unfused(cols, be, ga)
        assert got == [_lib.limbs_to_ints(t.cpu().numpy().view(np.uint64)) for t in (un, ud)]
        dom = [int(d) for d in pp.domain]
        want_n = [(int(a[i]) + be * dom[i] + ga) * (int(b[i]) + be * 2 * dom[i] + ga) * (int(c[i]) + be * 3 * dom[i] + ga) % R for i in range(n)]
        want_d = [(int(a[i]) + be * int(s1[i]) + ga) * (int(b[i]) + be * int(s2[i]) + ga) * (int(c[i]) + be * int(s3[i]) + ga) % R for i in range(n)]
        assert got == [want_n, want_d]
    # a length that is not a multiple of the workgroup size
    m = 37
    outs = [torch.zeros((m, 4), dtype=torch.int64, device="cuda") for _ in range(2)]
    ins = [t[:m].contiguous() for t in (cols[0], cols[1], cols[2], dev.evals["s_sigma1"], dev.evals["s_sigma2"], dev.evals["s_sigma3"], dev.ident)]
    plonk_perm_factors(outs[0].data_ptr(), outs[1].data_ptr(), [t.data_ptr() for t in ins], 99, 1234, m, torch.cuda.current_stream().cuda_stream)
    dom = [int(d) for d in pp.domain]
    assert _lib.limbs_to_ints(outs[0].cpu().numpy().view(np.uint64)) == [(int(a[i]) + 99 * dom[i] + 1234) * (int(b[i]) + 198 * dom[i] + 1234) * (int(c[i]) + 297 * dom[i] + 1234) % R for i in range(m)]


@pytest.mark.parametrize("zero_row", [0, 5, 15])
def test_device_grand_product_with_a_zero_denominator_equals_the_reference_loop(zero_row):
    """permutation.py:118-135 divides row by row and py_ecc's x / 0 is 0: a zero denominator zeroes z from the NEXT row on and keeps
    the rows before it.  The device computes z with scans and ONE inversion of the denominators' total; a zero total (which also
    covers the last row, that the reference never divides by) is detected and takes the row-exact path.  Challenges are chosen so
    that row `zero_row` has a zero denominator; z must equal the oracle's loop (oracle/plonk_ref.compute_accumulator) and the
    list prover's compute_accumulator."""
    import plonk_ref as pl
    from zkhip.plonk.permutation import compute_accumulator
    circuit, a, b, c = _chain_circuit(16, 9)
    srs = SRS.generate(16 + 80, seed=42)
    pp = preprocess(circuit, srs)
    n = pp.n
    assert n == 16
    dev = _device_from_circuit(circuit, pp, srs)
    s1, s2, s3 = build_permutation_polynomials(pp.sigma, n, pp.domain)
    beta = 0x1234567
    gamma = (-(int(a[zero_row]) + beta * int(s1[zero_row]))) % R          # a_i + beta * sigma_1(i) + gamma = 0
    want = pl.compute_accumulator([int(v) for v in a], [int(v) for v in b], [int(v) for v in c], list(pp.sigma), n, [int(d) for d in pp.domain], beta, gamma)
    # (a wired position shares its label with its partner, whose NUMERATOR factor vanishes as well: z may already be 0 one row earlier)
    assert want[0] == 1 and all(v == 0 for v in want[zero_row + 1:])
    assert [int(v) for v in compute_accumulator(a, b, c, pp.sigma, n, pp.domain, FR(beta), FR(gamma))] == want
    import torch
    cols = [torch.from_numpy(_limbs(col).view(np.int64)).cuda() for col in (a, b, c)]
    exact_calls = []
    exact = dev._accumulator_exact
    dev._accumulator_exact = lambda num, den: (exact_calls.append(1), exact(num, den))[1]
    got = dev._accumulator(cols, beta, gamma)
    assert exact_calls == [1]                                             # the zero total was seen and took the row-exact path
    assert _lib.limbs_to_ints(got.cpu().numpy().view(np.uint64)) == want
    # and with ordinary challenges the scan path gives the same z as the loop
    want2 = pl.compute_accumulator([int(v) for v in a], [int(v) for v in b], [int(v) for v in c], list(pp.sigma), n, [int(d) for d in pp.domain], 77, 99)
    assert _lib.limbs_to_ints(dev._accumulator(cols, 77, 99).cpu().numpy().view(np.uint64)) == want2
    assert exact_calls == [1]                                             # ... without the host path
