"""GPU tests (-m gpu) of the PLONK flow on the backend: preprocess
(8 iNTT + 8 commits), prove (5 rounds; quotient by coset NTT) and verify, restating the reference's
tests/plonk/test_circuit.py, test_crypto.py:308-456, test_prover.py and test_e2e.py:130-290.  The
deterministic part (preprocessing commitments) is pinned against the oracle; proofs are randomised in
the reference, so they are checked by verification, tampering and injected blinding."""
import copy

import pytest

import py_ref as o
from zkhip.field import FR, G1, ec_add, ec_mul
from zkhip.plonk.circuit import Circuit, Gate
from zkhip.plonk.permutation import K1, K2, build_permutation_polynomials, compute_accumulator
from zkhip.plonk.preprocessor import preprocess
from zkhip.plonk.prover import Proof, ProverState, prove, round1, round2, round3
from zkhip.plonk.srs import SRS
from zkhip.plonk.transcript import Transcript
from zkhip.plonk.verifier import verify

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def srs():
    return SRS.generate(20, seed=42)                     # tests/plonk/test_prover.py:43


@pytest.fixture(scope="module")
def toy(srs):
    circuit, a, b, c, pub = Circuit.x3_plus_x_plus_5_eq_35()
    pp = preprocess(circuit, srs)
    return circuit, a, b, c, pub, pp


def test_circuit_gates_and_permutation():
    circuit, a, b, c, pub = Circuit.x3_plus_x_plus_5_eq_35()
    assert circuit.n == 4 and pub == [FR(35)]
    assert all(g.check(a[i], b[i], c[i]) for i, g in enumerate(circuit.gates))
    assert not circuit.gates[0].check(3, 3, 10)
    sigma = circuit.build_copy_constraints()
    assert sorted(sigma) == list(range(12))              # a permutation of the 3n positions
    vals = a + b + c
    assert all(vals[i] == vals[sigma[i]] for i in range(12))   # cycles only link equal wires


def test_transcript_is_deterministic_and_binding():
    t1, t2 = Transcript(), Transcript()
    for t in (t1, t2):
        t.append_point(b"p", G1)
        t.append_scalar(b"s", FR(7))
    assert t1.challenge_scalar(b"c") == t2.challenge_scalar(b"c")
    assert t1.challenge_scalar(b"d") == t2.challenge_scalar(b"d")
    t3 = Transcript()
    t3.append_point(b"p", None)
    assert len(t3.state) == len(b"plonk") + 1 + 64
    t3.append_scalar(b"s", FR(7))
    assert t3.challenge_scalar(b"c") != Transcript().challenge_scalar(b"c")


def test_preprocess_matches_oracle(toy, srs):
    """Selector / permutation commitments are deterministic (tests/plonk/test_crypto.py:308-456)."""
    circuit, a, b, c, pub, pp = toy
    assert pp.n == 4 and pp.omega == FR(o.get_root_of_unity(4)) and len(pp.domain) == 4
    g1p = [(int(p[0]), int(p[1])) for p in srs.g1_powers]
    w = o.get_root_of_unity(4)
    sel = circuit.get_selector_polynomials()
    for name, evals in zip(("q_l", "q_r", "q_o", "q_m", "q_c"), sel):
        coeffs = o.ifft([int(v) for v in evals], w)
        assert [int(v) for v in getattr(pp, name + "_poly").coeffs] == o.trim(coeffs)
        exp = o.kzg_commit(coeffs, g1p)
        got = getattr(pp, name + "_comm")
        assert (None if got is None else (int(got[0]), int(got[1]))) == exp
    s_evals = build_permutation_polynomials(pp.sigma, 4, pp.domain)
    for k, evals in enumerate(s_evals, start=1):
        coeffs = o.ifft([int(v) for v in evals], w)
        got = getattr(pp, "s_sigma%d_comm" % k)
        assert (int(got[0]), int(got[1])) == o.kzg_commit(coeffs, g1p)
    # identity permutation check: S_sigma evaluations are a rearrangement of the coset labels
    labels = sorted(int(v) for col in (pp.domain, [K1 * d for d in pp.domain], [K2 * d for d in pp.domain]) for v in col)
    assert sorted(int(v) for col in s_evals for v in col) == labels


def test_accumulator_closes(toy):
    """z(omega^n) returns to 1: the grand product over a satisfied permutation is 1 (test_prover.py round 2)."""
    circuit, a, b, c, pub, pp = toy
    beta, gamma = FR(11), FR(13)
    z = compute_accumulator(a, b, c, pp.sigma, 4, pp.domain, beta, gamma)
    assert z[0] == FR(1) and len(z) == 4
    s1, s2, s3 = build_permutation_polynomials(pp.sigma, 4, pp.domain)
    i = 3
    num = (a[i] + beta * pp.domain[i] + gamma) * (b[i] + beta * K1 * pp.domain[i] + gamma) * (c[i] + beta * K2 * pp.domain[i] + gamma)
    den = (a[i] + beta * s1[i] + gamma) * (b[i] + beta * s2[i] + gamma) * (c[i] + beta * s3[i] + gamma)
    assert z[3] * num / den == FR(1)


def test_prove_and_verify_toy(toy, srs):
    circuit, a, b, c, pub, pp = toy
    proof = prove(circuit, a, b, c, pub, pp, srs)
    assert isinstance(proof, Proof)
    for f in Proof.FIELDS:
        assert getattr(proof, f) is not None
    assert isinstance(proof.a_comm, tuple) and len(proof.a_comm) == 2 and isinstance(proof.a_eval, FR)
    assert verify(proof, pub, pp, srs) is True
    assert verify(proof, [FR(36)], pp, srs) is True      # reference quirk: PI(x) = 0, public_inputs ignored (test_e2e.py:257-290)
    other = prove(circuit, a, b, c, pub, pp, srs)
    assert other.a_comm != proof.a_comm                  # random blinding (test_prover.py:710-722)
    assert verify(other, pub, pp, srs) is True


def test_injected_blinding_is_deterministic(toy, srs):
    circuit, a, b, c, pub, pp = toy
    bl = list(range(101, 110))
    p1 = prove(circuit, a, b, c, pub, pp, srs, blinding=bl)
    p2 = prove(circuit, a, b, c, pub, pp, srs, blinding=bl)
    assert all(getattr(p1, f) == getattr(p2, f) for f in Proof.FIELDS)
    assert verify(p1, pub, pp, srs)
    p0 = prove(circuit, a, b, c, pub, pp, srs, blinding=[0] * 9)   # unblinded: a(x) interpolates the wire values
    g1p = [(int(p[0]), int(p[1])) for p in srs.g1_powers]
    exp = o.kzg_commit(o.ifft([int(v) for v in a], o.get_root_of_unity(4)), g1p)
    assert (int(p0.a_comm[0]), int(p0.a_comm[1])) == exp and verify(p0, pub, pp, srs)


def test_soundness_by_tampering_every_field(toy, srs):
    """tests/plonk/test_e2e.py:198-254."""
    circuit, a, b, c, pub, pp = toy
    proof = prove(circuit, a, b, c, pub, pp, srs)
    for f in Proof.FIELDS:
        bad = copy.copy(proof)
        val = getattr(proof, f)
        setattr(bad, f, ec_add(val, G1) if isinstance(val, tuple) else val + FR(1))
        assert verify(bad, pub, pp, srs) is False, f


def test_inconsistent_witness_is_rejected(toy, srs):
    circuit, a, b, c, pub, pp = toy
    bad_c = list(c)
    bad_c[1] = bad_c[1] + FR(1)
    with pytest.raises(ValueError):
        prove(circuit, a, b, bad_c, pub, pp, srs)


def test_round_by_round_state(toy, srs):
    circuit, a, b, c, pub, pp = toy
    st = ProverState(a, b, c, pub, pp, srs)
    round1.execute(st)
    assert st.a_poly.degree == 5 and st.proof.a_comm is not None      # n + 1 with two blinding scalars
    for i in range(4):
        assert st.a_poly.evaluate(pp.domain[i]) == a[i]                # blinding vanishes on the domain
    round2.execute(st)
    assert st.z_poly.degree == 6 and st.z_poly.evaluate(FR(1)) == FR(1)
    round3.execute(st)
    zeta = FR(123456789)
    t_at = st.t_lo_poly.evaluate(zeta) + zeta ** 4 * st.t_mid_poly.evaluate(zeta) + zeta ** 8 * st.t_hi_poly.evaluate(zeta)
    # t * Z_H == full constraint at a random point
    al, be, ga = st.alpha, st.beta, st.gamma
    A, B, C, Z = (p.evaluate(zeta) for p in (st.a_poly, st.b_poly, st.c_poly, st.z_poly))
    gate = (pp.q_l_poly.evaluate(zeta) * A + pp.q_r_poly.evaluate(zeta) * B + pp.q_o_poly.evaluate(zeta) * C
            + pp.q_m_poly.evaluate(zeta) * A * B + pp.q_c_poly.evaluate(zeta))
    num = (A + be * zeta + ga) * (B + be * K1 * zeta + ga) * (C + be * K2 * zeta + ga) * Z
    den = ((A + be * pp.s_sigma1_poly.evaluate(zeta) + ga) * (B + be * pp.s_sigma2_poly.evaluate(zeta) + ga)
           * (C + be * pp.s_sigma3_poly.evaluate(zeta) + ga) * st.z_poly.evaluate(zeta * pp.omega))
    zh = zeta ** 4 - FR(1)
    l1 = zh / (FR(4) * (zeta - FR(1)))
    assert t_at * zh == gate + al * (num - den) + al * al * (Z - FR(1)) * l1


def test_other_circuits(srs):
    """A 1-gate circuit (n = 1) and a 6-gate chain padded to n = 8 (test_e2e.py's additional circuits)."""
    c1 = Circuit()
    c1.add_multiplication_gate()
    pp1 = preprocess(c1, srs)
    pr1 = prove(c1, [FR(6)], [FR(7)], [FR(42)], [], pp1, srs)
    assert verify(pr1, [], pp1, srs)
    c2 = Circuit()
    a, b, c = [], [], []
    x = FR(2)
    for i in range(6):                                   # x_{i+1} = x_i * x_i, chained by copy constraints
        c2.add_multiplication_gate()
        a.append(x); b.append(x); c.append(x * x)
        c2.add_copy_constraint(i, 0, i, 1)
        if i:
            c2.add_copy_constraint(i - 1, 2, i, 0)
        x = x * x
    pp2 = preprocess(c2, srs)
    assert pp2.n == 8 and c2.n == 8                      # padded in place, like the reference
    pr2 = prove(c2, a, b, c, [], pp2, srs)
    assert verify(pr2, [], pp2, srs)
    bad = copy.copy(pr2)
    bad.z_omega_eval = bad.z_omega_eval + FR(1)
    assert not verify(bad, [], pp2, srs)
