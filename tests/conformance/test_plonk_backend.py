"""Conformance of the backend-facing PLONK interface with what the reference's own test-suite expects (-m gpu):
tests/plonk/test_foundation.py (EC wrappers, fft / ifft / coset transforms, public-input polynomial),
tests/plonk/test_crypto.py (SRS, KZG commit / linearity / openings, preprocessor),
tests/plonk/test_prover.py (Proof / ProverState, the five rounds one by one, prove, invalid witness) and
tests/plonk/test_e2e.py (three circuits end to end, the 16 tamperings, public-input handling, cross-circuit soundness).
Same inputs and expectations as the reference's cases, restated as tables; every group operation, transform and
commitment below runs through libzkhip on the GPU."""
import copy
import random

import pytest

from zkhip.field import CURVE_ORDER, FQ, FQ12, FR, G1, G2, Z1, ec_add, ec_mul, ec_neg, ec_pairing, get_root_of_unity
from zkhip.plonk.circuit import Circuit
from zkhip.plonk.kzg import commit, create_witness, verify_opening
from zkhip.plonk.permutation import K1, K2
from zkhip.plonk.polynomial import Polynomial, fft, ifft
from zkhip.plonk.preprocessor import PreprocessedData, preprocess
from zkhip.plonk.prover import Proof, ProverState, prove, round1, round2, round3, round4, round5
from zkhip.plonk.srs import SRS
from zkhip.plonk.transcript import Transcript
from zkhip.plonk.utils import coset_fft, coset_ifft, lagrange_basis_eval, public_input_poly_eval, public_input_polynomial, vanishing_poly_eval
from zkhip.plonk.verifier import verify

pytestmark = pytest.mark.gpu
R = CURVE_ORDER


def on_curve(pt):
    """bn128.is_on_curve(pt, b) for a G1 point (y^2 = x^3 + 3; infinity passes)."""
    return pt is None or pt[1] * pt[1] == pt[0] * pt[0] * pt[0] + FQ(3)


# ------------------------------------------------------------------ EC wrappers (test_foundation.py:106-174)
def test_ec_wrappers():
    assert ec_mul(G1, 1) == G1
    assert ec_mul(G1, 0) is None and ec_mul(G1, R) is None
    assert ec_mul(G1, FR(5)) == ec_mul(G1, 5)
    assert ec_add(G1, Z1) == G1
    assert ec_add(G1, G1) == ec_mul(G1, 2)
    p3, p7 = ec_mul(G1, 3), ec_mul(G1, 7)
    assert ec_add(p3, p7) == ec_add(p7, p3) == ec_mul(G1, 10)
    a, b, c = ec_mul(G1, 2), ec_mul(G1, 3), ec_mul(G1, 5)
    assert ec_add(ec_add(a, b), c) == ec_add(a, ec_add(b, c))
    p5 = ec_mul(G1, 5)
    assert ec_add(p5, ec_neg(p5)) is None
    e1 = ec_pairing(G2, G1)
    assert e1 is not None and e1 != ec_pairing(G2, Z1) and ec_pairing(G2, Z1) == FQ12.one()
    assert ec_pairing(G2, ec_mul(G1, 15)) == ec_pairing(ec_mul(G2, 5), ec_mul(G1, 3))


# ------------------------------------------------------------------ transforms (test_foundation.py:465-545, 673-765)
def test_fft_ifft_and_interpolation():
    assert fft([FR(7)], FR(1)) == [FR(7)] and ifft([FR(7)], FR(1)) == [FR(7)]
    w4, w8 = get_root_of_unity(4), get_root_of_unity(8)
    coeffs = [FR(1), FR(2), FR(3), FR(4)]
    ev = fft(coeffs, w4)
    assert len(ev) == 4 and ev[0] == Polynomial(coeffs).evaluate(FR(1)) == FR(10)
    c8 = [FR(i) for i in range(8)]
    p8 = Polynomial(c8)
    assert fft(c8, w8) == [p8.evaluate(w8 ** i) for i in range(8)]
    evals = [FR(10), FR(5), FR(3), FR(7)]
    assert fft(ifft(evals, w4), w4) == evals
    orig = [FR(i * 3 + 1) for i in range(8)]
    assert ifft(fft(orig, w8), w8) == orig
    orig = [FR(7), FR(11), FR(13), FR(17)]
    assert fft(ifft(orig, w4), w4) == orig
    p = Polynomial([FR(1), FR(2), FR(3), FR(0)])
    assert Polynomial.from_evaluations([p.evaluate(w4 ** i) for i in range(4)], w4) == p
    assert Polynomial.from_evaluations([FR(5)] * 4, w4) == Polynomial([FR(5)])


def test_coset_transforms():
    w, k = get_root_of_unity(4), FR(5)
    coeffs = [FR(1), FR(2), FR(3), FR(0)]
    p = Polynomial(coeffs)
    assert coset_fft(coeffs, w, k) == [p.evaluate(k * w ** i) for i in range(4)]
    orig = [FR(7), FR(11), FR(13), FR(17)]
    assert coset_ifft(coset_fft(orig, w, k), w, k) == orig
    c = [FR(1), FR(2), FR(3), FR(4)]
    assert coset_fft(c, w, FR(5)) == coset_fft(c, w) and coset_ifft(c, w, FR(5)) == coset_ifft(c, w)


def test_public_input_polynomial():
    n, w = 4, get_root_of_unity(4)
    assert public_input_polynomial([], n, w).is_zero()
    pi = public_input_polynomial([FR(7)], n, w)
    assert [pi.evaluate(w ** i) for i in range(n)] == [FR(7), FR(0), FR(0), FR(0)]
    pi = public_input_polynomial([FR(10), FR(20)], n, w)
    assert [pi.evaluate(w ** i) for i in range(n)] == [FR(10), FR(20), FR(0), FR(0)]
    pub = [FR(5), FR(10)]
    assert public_input_poly_eval(pub, n, w, FR(42)) == public_input_polynomial(pub, n, w).evaluate(FR(42))


# ------------------------------------------------------------------ SRS (test_crypto.py:45-125)
@pytest.fixture(scope="module")
def srs_small():
    return SRS.generate(max_degree=8, seed=42)


def test_srs(srs_small):
    s = srs_small
    assert len(s.g1_powers) == 9 and len(s.g2_powers) == 2 and s.max_degree == 8
    assert s.g1_powers[0] == G1 and s.g2_powers[0] == G2
    assert all(pt is not None for pt in s.g1_powers) and all(pt is not None for pt in s.g2_powers)
    assert all(s.g1_powers[i] != s.g1_powers[i + 1] for i in range(8))
    a, b = SRS.generate(max_degree=4, seed=99), SRS.generate(max_degree=4, seed=99)
    assert a.g1_powers == b.g1_powers and a.g2_powers == b.g2_powers
    assert SRS.generate(max_degree=4, seed=1).g1_powers != SRS.generate(max_degree=4, seed=2).g1_powers
    r = SRS.generate(max_degree=2)                                            # random tau
    assert len(r.g1_powers) == 3 and len(r.g2_powers) == 2
    z = SRS.generate(max_degree=0, seed=42)
    assert z.g1_powers == [G1] and z.max_degree == 0


# ------------------------------------------------------------------ KZG (test_crypto.py:130-330)
def test_kzg_commit(srs_small):
    s = srs_small
    assert commit(Polynomial([FR(7)]), s) == ec_mul(G1, FR(7))
    assert commit(Polynomial([FR(3), FR(5)]), s) == ec_add(ec_mul(s.g1_powers[0], FR(3)), ec_mul(s.g1_powers[1], FR(5)))
    assert commit(Polynomial([FR(0)]), s) is None
    with pytest.raises(ValueError):
        commit(Polynomial([FR(1)] * 10), s)                                   # degree 9 > max_degree 8
    assert commit(Polynomial([FR(0)] * 8 + [FR(1)]), s) is not None           # exactly max_degree
    poly = Polynomial([FR(2), FR(3)])
    assert commit(poly * FR(5), s) == ec_mul(commit(poly, s), FR(5))
    p, q = Polynomial([FR(1), FR(2)]), Polynomial([FR(3), FR(4)])
    assert commit(p + q, s) == ec_add(commit(p, s), commit(q, s))
    p, q = Polynomial([FR(1)]), Polynomial([FR(0), FR(0), FR(5)])
    assert commit(p + q, s) == ec_add(commit(p, s), commit(q, s))
    p = Polynomial([FR(3), FR(7)])
    assert commit(p + Polynomial([FR(0)]), s) == commit(p, s)


@pytest.mark.parametrize("coeffs,point,value", [
    ([5], FR(7), 5), ([1, 2], FR(3), 7), ([1, 1, 1], FR(2), 7), ([42, 3, 5], FR(0), 42), ([1] * 6, FR(2), 63), ([1, 2], 3, 7),
])
def test_kzg_valid_openings(srs_small, coeffs, point, value):
    poly = Polynomial(coeffs)
    assert poly.evaluate(point) == FR(value)
    C, proof = commit(poly, srs_small), create_witness(poly, point, srs_small)
    assert verify_opening(C, proof, point, FR(value), srs_small)


def test_kzg_openings_more(srs_small):
    s = srs_small
    poly = Polynomial([FR(1), FR(2)])
    C, proof = commit(poly, s), create_witness(poly, FR(3), s)
    assert not verify_opening(C, proof, FR(3), FR(8), s)                      # wrong evaluation
    assert not verify_opening(C, proof, FR(5), FR(7), s)                      # proof for 3, claimed at 5
    other = commit(Polynomial([FR(3), FR(4)]), s)
    assert not verify_opening(other, create_witness(poly, FR(5), s), FR(5), poly.evaluate(FR(5)), s)
    quad = Polynomial([FR(2), FR(3), FR(1)])
    Cq = commit(quad, s)
    for z in (FR(0), FR(1), FR(5), FR(100)):
        assert verify_opening(Cq, create_witness(quad, z, s), z, quad.evaluate(z), s)


# ------------------------------------------------------------------ preprocessor (test_crypto.py:335-504)
@pytest.fixture(scope="module")
def toy_pp():
    circuit, a, b, c, pub = Circuit.x3_plus_x_plus_5_eq_35()
    srs = SRS.generate(max_degree=32, seed=1234)
    return circuit, srs, preprocess(circuit, srs)


def test_preprocess(toy_pp):
    circuit, srs, pp = toy_pp
    assert isinstance(pp, PreprocessedData)
    assert pp.n >= 4 and pp.n & (pp.n - 1) == 0
    assert pp.omega ** pp.n == FR(1)
    power = FR(1)
    for k in range(1, pp.n):
        power = power * pp.omega
        assert power != FR(1)
    assert len(pp.domain) == pp.n and all(pp.domain[i] == pp.omega ** i for i in range(pp.n))
    names = ("q_l", "q_r", "q_o", "q_m", "q_c", "s_sigma1", "s_sigma2", "s_sigma3")
    for name in names:
        poly = getattr(pp, name + "_poly")
        assert isinstance(poly, Polynomial) and getattr(pp, name + "_comm") == commit(poly, srs)
    assert pp.q_l_comm is not None and pp.q_o_comm is not None and pp.q_m_comm is not None
    assert pp.s_sigma1_comm is not None and pp.s_sigma2_comm is not None and pp.s_sigma3_comm is not None
    assert len(pp.sigma) == 3 * pp.n and sorted(pp.sigma) == list(range(3 * pp.n))
    assert pp.num_public_inputs == 1
    sel = circuit.get_selector_polynomials()
    for name, col in zip(names[:5], sel):                                      # the selector polynomials interpolate the gate table
        assert [getattr(pp, name + "_poly").evaluate(d) for d in pp.domain] == list(col)
    coset = {int(k * d) for k in (FR(1), K1, K2) for d in pp.domain}
    for name in names[5:]:
        assert all(int(getattr(pp, name + "_poly").evaluate(d)) in coset for d in pp.domain)


def test_preprocess_pads_to_power_of_two():
    c = Circuit()
    for _ in range(3):
        c.add_addition_gate()
    pp = preprocess(c, SRS.generate(max_degree=16, seed=5))
    assert pp.n == 4 and c.n == 4 and c.gates[3].check(FR(1), FR(2), FR(99))   # padded in place with an all-zero gate


# ------------------------------------------------------------------ prover rounds (test_prover.py)
@pytest.fixture(scope="module")
def setup():
    circuit, a, b, c, pub = Circuit.x3_plus_x_plus_5_eq_35()
    srs = SRS.generate(20, seed=42)
    return circuit, a, b, c, pub, srs, preprocess(circuit, srs)


def _state(setup, upto):
    circuit, a, b, c, pub, srs, pp = setup
    st = ProverState(a, b, c, pub, pp, srs)
    for rnd in (round1, round2, round3, round4, round5)[:upto]:
        rnd.execute(st)
    return st


def test_proof_and_state_init(setup):
    circuit, a, b, c, pub, srs, pp = setup
    proof = Proof()
    assert all(getattr(proof, f) is None for f in (
        "a_comm", "b_comm", "c_comm", "z_comm", "t_lo_comm", "t_mid_comm", "t_hi_comm", "a_eval", "b_eval", "c_eval", "s_sigma1_eval",
        "s_sigma2_eval", "z_omega_eval", "r_eval", "W_zeta_comm", "W_zeta_omega_comm"))
    st = ProverState(a, b, c, pub, pp, srs)
    assert (st.a_vals, st.b_vals, st.c_vals, st.public_inputs) == (a, b, c, pub)
    assert st.preprocessed is pp and st.srs is srs and (st.n, st.omega, st.domain) == (pp.n, pp.omega, pp.domain)
    assert all(getattr(st, f) is None for f in ("a_poly", "b_poly", "c_poly", "z_poly", "t_lo_poly", "t_mid_poly", "t_hi_poly", "beta",
                                                "gamma", "alpha", "zeta", "v"))
    assert isinstance(st.proof, Proof) and isinstance(st.transcript, Transcript) and st.build_proof() is st.proof


def test_round1(setup):
    st = _state(setup, 1)
    srs, n = setup[5], st.n
    for name in "abc":
        poly, comm = getattr(st, name + "_poly"), getattr(st.proof, name + "_comm")
        assert isinstance(poly, Polynomial) and poly.degree >= n
        assert [poly.evaluate(d) for d in st.domain] == getattr(st, name + "_vals")   # blinding vanishes on the domain
        assert comm is not None and on_curve(comm) and comm == commit(poly, srs)
    assert st.pi_poly is not None and st.pi_poly.is_zero()
    assert isinstance(st.a_poly.evaluate(FR(123456789)), FR)
    assert _state(setup, 1).a_poly.coeffs != st.a_poly.coeffs                     # fresh blinding every run


def test_round2(setup):
    st = _state(setup, 2)
    assert isinstance(st.beta, FR) and isinstance(st.gamma, FR) and st.beta != FR(0) and st.gamma != FR(0) and st.beta != st.gamma
    assert isinstance(st.z_poly, Polynomial) and st.z_poly.degree >= st.n
    assert st.z_poly.evaluate(st.domain[0]) == FR(1)
    assert all(isinstance(st.z_poly.evaluate(d), FR) for d in st.domain)
    assert st.proof.z_comm is not None and on_curve(st.proof.z_comm) and st.proof.z_comm == commit(st.z_poly, setup[5])


def test_round3(setup):
    st = _state(setup, 3)
    srs, n, pp = setup[5], st.n, st.preprocessed
    assert isinstance(st.alpha, FR) and st.alpha != FR(0)
    for name in ("t_lo", "t_mid", "t_hi"):
        poly, comm = getattr(st, name + "_poly"), getattr(st.proof, name + "_comm")
        assert isinstance(poly, Polynomial) and comm is not None and on_curve(comm) and comm == commit(poly, srs)
    assert st.t_lo_poly.degree < n and st.t_mid_poly.degree < n
    # t(x) Z_H(x) equals the constraint polynomial at points off the domain (the reference's test only checks the types here)
    for x in (FR(11), FR(37), FR(7777), FR(9999)):
        xn = x ** n
        t_x = st.t_lo_poly.evaluate(x) + xn * st.t_mid_poly.evaluate(x) + xn * xn * st.t_hi_poly.evaluate(x)
        a, b, c, z = (p.evaluate(x) for p in (st.a_poly, st.b_poly, st.c_poly, st.z_poly))
        zw = st.z_poly.evaluate(x * st.omega)
        gate = pp.q_l_poly.evaluate(x) * a + pp.q_r_poly.evaluate(x) * b + pp.q_o_poly.evaluate(x) * c + pp.q_m_poly.evaluate(x) * a * b + pp.q_c_poly.evaluate(x)
        num = (a + st.beta * x + st.gamma) * (b + st.beta * K1 * x + st.gamma) * (c + st.beta * K2 * x + st.gamma) * z
        den = ((a + st.beta * pp.s_sigma1_poly.evaluate(x) + st.gamma) * (b + st.beta * pp.s_sigma2_poly.evaluate(x) + st.gamma)
               * (c + st.beta * pp.s_sigma3_poly.evaluate(x) + st.gamma) * zw)
        l1 = lagrange_basis_eval(0, n, st.omega, x)
        assert gate + st.alpha * (num - den) + st.alpha * st.alpha * (z - FR(1)) * l1 == t_x * vanishing_poly_eval(n, x)


def test_round4(setup):
    st = _state(setup, 4)
    pr, pp = st.proof, st.preprocessed
    assert isinstance(st.zeta, FR) and st.zeta != FR(0)
    want = {"a_eval": st.a_poly.evaluate(st.zeta), "b_eval": st.b_poly.evaluate(st.zeta), "c_eval": st.c_poly.evaluate(st.zeta),
            "s_sigma1_eval": pp.s_sigma1_poly.evaluate(st.zeta), "s_sigma2_eval": pp.s_sigma2_poly.evaluate(st.zeta),
            "z_omega_eval": st.z_poly.evaluate(st.zeta * st.omega)}
    for name, val in want.items():
        assert isinstance(getattr(pr, name), FR) and getattr(pr, name) == val


def test_round5(setup):
    st = _state(setup, 5)
    pr, pp, n, zeta = st.proof, st.preprocessed, st.n, st.zeta
    assert isinstance(st.v, FR) and st.v != FR(0) and isinstance(pr.r_eval, FR)
    assert pr.W_zeta_comm is not None and pr.W_zeta_omega_comm is not None and on_curve(pr.W_zeta_comm) and on_curve(pr.W_zeta_omega_comm)
    zn = zeta ** n
    t_eval = st.t_lo_poly.evaluate(zeta) + zn * st.t_mid_poly.evaluate(zeta) + zn * zn * st.t_hi_poly.evaluate(zeta)
    assert pr.r_eval == t_eval * vanishing_poly_eval(n, zeta)                      # r(zeta) = t(zeta) Z_H(zeta)
    a, b, c, s1, s2, zw = pr.a_eval, pr.b_eval, pr.c_eval, pr.s_sigma1_eval, pr.s_sigma2_eval, pr.z_omega_eval
    alpha, beta, gamma = st.alpha, st.beta, st.gamma
    l1 = lagrange_basis_eval(0, n, st.omega, zeta)
    gate = (pp.q_m_poly.evaluate(zeta) * a * b + pp.q_l_poly.evaluate(zeta) * a + pp.q_r_poly.evaluate(zeta) * b + pp.q_o_poly.evaluate(zeta) * c
            + pp.q_c_poly.evaluate(zeta) + st.pi_poly.evaluate(zeta))
    z_zeta = st.z_poly.evaluate(zeta)
    ab = (a + beta * s1 + gamma) * (b + beta * s2 + gamma)
    perm = (alpha * (a + beta * zeta + gamma) * (b + beta * K1 * zeta + gamma) * (c + beta * K2 * zeta + gamma) * z_zeta
            - alpha * ab * beta * zw * pp.s_sigma3_poly.evaluate(zeta) + (FR(0) - alpha * ab * zw * (c + gamma)))
    boundary = alpha * alpha * l1 * z_zeta + (FR(0) - alpha * alpha * l1)
    assert pr.r_eval == gate + perm + boundary


def test_prove_and_invalid_witness(setup):
    circuit, a, b, c, pub, srs, pp = setup
    p1, p2 = prove(circuit, a, b, c, pub, pp, srs), prove(circuit, a, b, c, pub, pp, srs)
    assert isinstance(p1, Proof) and isinstance(p2, Proof) and p1.a_comm != p2.a_comm
    for f in Proof.FIELDS:
        assert getattr(p1, f) is not None
    assert all(on_curve(getattr(p1, f)) for f in Proof.FIELDS if f.endswith("_comm"))
    assert all(isinstance(getattr(p1, f), FR) for f in Proof.FIELDS if f.endswith("_eval"))
    st = ProverState(a, b, [FR(10), FR(27), FR(30), FR(35)], [FR(35)], pp, srs)    # c_0 = 10 instead of 9
    round1.execute(st)
    round2.execute(st)
    with pytest.raises(ValueError, match="나누어 떨어지지 않"):                      # test_prover.py:746
        round3.execute(st)


# ------------------------------------------------------------------ end to end (test_e2e.py)
def _pipeline(builder, seed):
    circuit, a, b, c, pub = builder()
    srs = SRS.generate(max_degree=3 * circuit.n + 10, seed=seed)
    pp = preprocess(circuit, srs)
    return dict(circuit=circuit, a=a, b=b, c=c, pub=pub, srs=srs, pp=pp, proof=prove(circuit, a, b, c, pub, pp, srs))


def _one_gate(kind, a, b, c):
    def build():
        circ = Circuit()
        getattr(circ, kind)()
        return circ, [FR(a)], [FR(b)], [FR(c)], []
    return build


@pytest.fixture(scope="module")
def x3():
    return _pipeline(Circuit.x3_plus_x_plus_5_eq_35, 12345)


@pytest.fixture(scope="module")
def add1():
    return _pipeline(_one_gate("add_addition_gate", 3, 7, 10), 9999)             # n = 1


@pytest.fixture(scope="module")
def mul1():
    return _pipeline(_one_gate("add_multiplication_gate", 4, 5, 20), 7777)       # n = 1


def test_e2e_three_circuits_verify(x3, add1, mul1):
    for d in (x3, add1, mul1):
        assert verify(d["proof"], d["pub"], d["pp"], d["srs"]) is True
        assert all(g.check(d["a"][i], d["b"][i], d["c"][i]) for i, g in enumerate(d["circuit"].gates) if i < len(d["a"]))
    assert all(getattr(x3["proof"], f) is not None for f in Proof.FIELDS)
    srs = SRS.generate(max_degree=3 * 4 + 10, seed=12345)                          # fresh randomness: still accepted
    circuit, a, b, c, pub = Circuit.x3_plus_x_plus_5_eq_35()
    pp = preprocess(circuit, srs)
    assert verify(prove(circuit, a, b, c, pub, pp, srs), pub, pp, srs) is True


SCALARS = ["a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval", "r_eval"]
POINTS = ["a_comm", "b_comm", "c_comm", "z_comm", "t_lo_comm", "t_mid_comm", "t_hi_comm", "W_zeta_comm", "W_zeta_omega_comm"]


@pytest.mark.parametrize("field", SCALARS + POINTS)
def test_e2e_single_field_tampering_is_rejected(x3, field):
    bad = copy.deepcopy(x3["proof"])
    if field in SCALARS:
        setattr(bad, field, getattr(bad, field) + FR(1))
    else:
        setattr(bad, field, ec_mul(G1, FR(random.randint(1, R - 1))))
    assert verify(bad, x3["pub"], x3["pp"], x3["srs"]) is False


def test_e2e_public_inputs_cross_circuit_and_double_tampering(x3, add1):
    # PI(x) = 0 in the reference: the public_inputs argument does not enter the check (test_e2e.py:262-288)
    assert verify(x3["proof"], [FR(999)], x3["pp"], x3["srs"]) is True
    assert verify(x3["proof"], [], x3["pp"], x3["srs"]) is True
    assert verify(x3["proof"], x3["pub"], add1["pp"], add1["srs"]) is False
    assert verify(add1["proof"], add1["pub"], x3["pp"], x3["srs"]) is False
    bad = copy.deepcopy(x3["proof"])
    bad.a_eval, bad.b_eval = bad.a_eval + FR(1), bad.b_eval + FR(1)
    assert verify(bad, x3["pub"], x3["pp"], x3["srs"]) is False
    bad = copy.deepcopy(x3["proof"])
    bad.a_comm, bad.a_eval = ec_mul(G1, FR(424242)), bad.a_eval + FR(1)
    assert verify(bad, x3["pub"], x3["pp"], x3["srs"]) is False


# ------------------------------------------------------------------ round-by-round resume through the wire format
def test_prover_resumes_from_persisted_state_between_rounds(setup):
    """The reference's web flow runs ONE round per request: it persists the round's outputs as JSON and rebuilds a fresh
    ProverState from storage before the next round (plonk_routes.py:298-373, the checkpoint / resume of SURVEY.md section 5).
    The same walk through zkhip.serializers -- every object JSON-encoded and decoded between rounds, a new ProverState each
    time -- must end in the proof an uninterrupted run gives for the same blinding scalars, and that proof must survive its own
    wire format and verify."""
    import json
    from zkhip import serializers as ser
    circuit, a, b, c, pub, srs, pp = setup
    blinding = [123456789 + 1000003 * k for k in range(9)]
    want = prove(circuit, a, b, c, pub, pp, srs, blinding=blinding)

    store = json.loads(json.dumps({                                           # what the routes keep in their database
        "srs": ser.serialize_srs(srs), "pp": ser.serialize_preprocessed(pp),
        "a": ser.serialize_fr_list(a), "b": ser.serialize_fr_list(b), "c": ser.serialize_fr_list(c), "pub": ser.serialize_fr_list(pub)}))

    def rebuild(upto, left):
        srs2, pp2 = ser.deserialize_srs(store["srs"]), ser.deserialize_preprocessed(store["pp"])
        st = ProverState(ser.deserialize_fr_list(store["a"]), ser.deserialize_fr_list(store["b"]), ser.deserialize_fr_list(store["c"]),
                         ser.deserialize_fr_list(store["pub"]), pp2, srs2, blinding=left)
        if upto >= 2:
            r1 = store["r1"]
            st.a_poly, st.b_poly, st.c_poly, st.pi_poly = (ser.deserialize_poly(r1[k]) for k in ("a_poly", "b_poly", "c_poly", "pi_poly"))
            st.proof.a_comm, st.proof.b_comm, st.proof.c_comm = (ser.deserialize_g1(r1[k]) for k in ("a_comm", "b_comm", "c_comm"))
            st.transcript = ser.deserialize_transcript(r1["transcript"])
        if upto >= 3:
            r2 = store["r2"]
            st.beta, st.gamma, st.z_poly = ser.deserialize_fr(r2["beta"]), ser.deserialize_fr(r2["gamma"]), ser.deserialize_poly(r2["z_poly"])
            st.proof.z_comm = ser.deserialize_g1(r2["z_comm"])
            st.transcript = ser.deserialize_transcript(r2["transcript"])
        if upto >= 4:
            r3 = store["r3"]
            st.alpha = ser.deserialize_fr(r3["alpha"])
            st.t_lo_poly, st.t_mid_poly, st.t_hi_poly = (ser.deserialize_poly(r3[k]) for k in ("t_lo_poly", "t_mid_poly", "t_hi_poly"))
            st.proof.t_lo_comm, st.proof.t_mid_comm, st.proof.t_hi_comm = (ser.deserialize_g1(r3[k]) for k in ("t_lo_comm", "t_mid_comm", "t_hi_comm"))
            st.transcript = ser.deserialize_transcript(r3["transcript"])
        if upto >= 5:
            r4 = store["r4"]
            st.zeta = ser.deserialize_fr(r4["zeta"])
            for k in ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval"):
                setattr(st.proof, k, ser.deserialize_fr(r4[k]))
            st.transcript = ser.deserialize_transcript(r4["transcript"])
        return st

    def persist(key, obj):
        store[key] = json.loads(json.dumps(obj))

    st = rebuild(1, blinding[:6])
    round1.execute(st)
    persist("r1", {**{k: ser.serialize_poly(getattr(st, k)) for k in ("a_poly", "b_poly", "c_poly", "pi_poly")},
                   **{k: ser.serialize_g1(getattr(st.proof, k)) for k in ("a_comm", "b_comm", "c_comm")},
                   "transcript": ser.serialize_transcript(st.transcript)})
    st = rebuild(2, blinding[6:])
    round2.execute(st)
    persist("r2", {"beta": ser.serialize_fr(st.beta), "gamma": ser.serialize_fr(st.gamma), "z_poly": ser.serialize_poly(st.z_poly),
                   "z_comm": ser.serialize_g1(st.proof.z_comm), "transcript": ser.serialize_transcript(st.transcript)})
    st = rebuild(3, [])
    round3.execute(st)
    persist("r3", {"alpha": ser.serialize_fr(st.alpha), **{k: ser.serialize_poly(getattr(st, k)) for k in ("t_lo_poly", "t_mid_poly", "t_hi_poly")},
                   **{k: ser.serialize_g1(getattr(st.proof, k)) for k in ("t_lo_comm", "t_mid_comm", "t_hi_comm")},
                   "transcript": ser.serialize_transcript(st.transcript)})
    st = rebuild(4, [])
    round4.execute(st)
    persist("r4", {"zeta": ser.serialize_fr(st.zeta), **{k: ser.serialize_fr(getattr(st.proof, k)) for k in
                                                          ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval")},
                   "transcript": ser.serialize_transcript(st.transcript)})
    st = rebuild(5, [])
    round5.execute(st)
    got = st.build_proof()
    for f in Proof.FIELDS:
        assert getattr(got, f) == getattr(want, f), f
    back = ser.deserialize_proof(json.loads(json.dumps(ser.serialize_proof(got))))
    assert all(getattr(back, f) == getattr(want, f) for f in Proof.FIELDS)
    assert verify(back, ser.deserialize_fr_list(store["pub"]), ser.deserialize_preprocessed(store["pp"]), ser.deserialize_srs(store["srs"])) is True
