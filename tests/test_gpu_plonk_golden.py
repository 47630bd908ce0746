"""GPU tests (-m gpu): the PLONK provers of the backend against the reference-shaped oracle.

tests/golden/plonk_proofs.json holds, for the reference's x^3+x+5=35 circuit (n = 4) and a six-gate circuit padded to n = 8, every
proof field computed the reference's way (oracle/plonk_ref.py: O(n^2) Polynomial.__mul__, poly_div by Z_H, Horner) with the nine
blinding scalars fixed.  With the same blinding injected
  * zkhip.plonk.prover.prove (rounds on the backend: inverse NTT, quotient by coset NTT, MSM commitments) and
  * zkhip.plonk.prover_device.DevicePlonk.prove (every vector resident in HBM, fused quotient kernel, scans)
must give the same 16 fields, challenge for challenge; the ORACLE's verifier (one ec_mul per term, py_ecc-shaped pairing) must
accept the GPU proof and reject its 16 single-field tamperings (tests/plonk/test_e2e.py:198-254), and the backend's verifier must
agree with the oracle's verdict on each of them."""
import copy
import json
import os

import pytest

import plonk_ref as pl
import py_ref as o
from test_plonk_oracle import load_case
from zkhip import _lib
from zkhip.field import FQ, FR, g1_to_limbs
from zkhip.plonk.circuit import Circuit, Gate
from zkhip.plonk.permutation import build_permutation_polynomials
from zkhip.plonk.preprocessor import preprocess
from zkhip.plonk.prover import ProverState, prove, round1, round2, round3, round4, round5
from zkhip.plonk.prover_device import DevicePlonk
from zkhip.plonk.srs import SRS
from zkhip.plonk.verifier import verify

pytestmark = pytest.mark.gpu
R = o.R


@pytest.fixture(scope="module")
def cases(golden_dir):
    with open(os.path.join(golden_dir, "plonk_proofs.json")) as f:
        return json.load(f)["cases"]


def _facade_circuit(case):
    c = Circuit()
    c.gates = [Gate(*[int(v) for v in g]) for g in case["gates"]]
    c.copy_constraints = [tuple(cc) for cc in case["copy_constraints"]]
    c.num_public_inputs = len(case["public_inputs"])
    return c


def _as_oracle_proof(p):
    """backend Proof (FQ / FR values) -> oracle Proof (ints)."""
    q = pl.Proof()
    for f in pl.PROOF_FIELDS:
        v = getattr(p, f)
        setattr(q, f, (None if v is None else (int(v[0]), int(v[1]))) if f in pl.PROOF_POINTS else int(v))
    return q


def _assert_fields(got, case):
    for f in pl.PROOF_FIELDS:
        want = case["proof"][f]
        v = getattr(got, f)
        if f in pl.PROOF_POINTS:
            assert (None if v is None else [str(int(v[0])), str(int(v[1]))]) == want, f
        else:
            assert str(int(v)) == want, f


@pytest.fixture(scope="module", params=["toy_x3", "six_gates_n8"])
def setup(request, cases):
    case = cases[request.param]
    srs = SRS.generate(case["srs"]["max_degree"], seed=case["srs"]["seed"])
    circuit = _facade_circuit(case)
    pp = preprocess(circuit, srs)
    cols = [[FR(int(v)) for v in case[k]] for k in ("a_vals", "b_vals", "c_vals")]
    return case, srs, circuit, pp, cols


def test_preprocessing_commitments_match_fixture(setup):
    case, srs, circuit, pp, cols = setup
    assert pp.n == case["n"] and pp.sigma == case["sigma"]
    for k, v in case["preprocessed"].items():
        got = getattr(pp, k)
        assert [str(int(got[0])), str(int(got[1]))] == v, k


def test_list_prover_round_by_round(setup):
    """Challenges and polynomials after every round against the oracle's (the transcript order and the blinding layout are
    what a shared prover/verifier slip would hide)."""
    case, srs, circuit, pp, cols = setup
    st = ProverState(cols[0], cols[1], cols[2], [FR(int(v)) for v in case["public_inputs"]], pp, srs, blinding=[int(v) for v in case["blinding"]])
    coeffs = lambda poly, key: [str(int(c)) for c in poly.coeffs] == case["polys"][key]
    round1.execute(st)
    assert coeffs(st.a_poly, "a_poly") and coeffs(st.b_poly, "b_poly") and coeffs(st.c_poly, "c_poly")
    round2.execute(st)
    assert str(int(st.beta)) == case["challenges"]["beta"] and str(int(st.gamma)) == case["challenges"]["gamma"]
    assert coeffs(st.z_poly, "z_poly")
    round3.execute(st)
    assert str(int(st.alpha)) == case["challenges"]["alpha"]
    assert coeffs(st.t_lo_poly, "t_lo_poly") and coeffs(st.t_mid_poly, "t_mid_poly") and coeffs(st.t_hi_poly, "t_hi_poly")
    round4.execute(st)
    assert str(int(st.zeta)) == case["challenges"]["zeta"]
    round5.execute(st)
    assert str(int(st.v)) == case["challenges"]["v"]
    _assert_fields(st.build_proof(), case)


def test_list_prover_matches_fixture_and_oracle_verifier(setup):
    case, srs, circuit, pp, cols = setup
    pub = [FR(int(v)) for v in case["public_inputs"]]
    got = prove(circuit, cols[0], cols[1], cols[2], pub, pp, srs, blinding=[int(v) for v in case["blinding"]])
    _assert_fields(got, case)
    assert verify(got, pub, pp, srs) is True
    # the oracle's verifier on the GPU proof, with the ORACLE's preprocessing and SRS
    ocirc, oa, ob, oc, opub, oblind, osrs, _ = load_case(case)
    opp = pl.preprocess(ocirc, osrs)
    oproof = _as_oracle_proof(got)
    assert pl.verify(oproof, opub, opp, osrs) is True
    if case["n"] != 4:
        return
    fake_o = o.ec_mul(o.G1, 0x1D0F4C0FFEE)
    fake = (FQ(fake_o[0]), FQ(fake_o[1]))
    for f in pl.PROOF_SCALARS:                                        # tests/plonk/test_e2e.py:205-222
        bad_o, bad = copy.copy(oproof), copy.copy(got)
        setattr(bad_o, f, (getattr(oproof, f) + 1) % R)
        setattr(bad, f, getattr(got, f) + FR(1))
        assert pl.verify(bad_o, opub, opp, osrs) is False, f
        assert verify(bad, pub, pp, srs) is False, f
    for f in pl.PROOF_POINTS:                                         # tests/plonk/test_e2e.py:234-254
        bad_o, bad = copy.copy(oproof), copy.copy(got)
        setattr(bad_o, f, fake_o)
        setattr(bad, f, fake)
        assert pl.verify(bad_o, opub, opp, osrs) is False, f
        assert verify(bad, pub, pp, srs) is False, f


def test_device_prover_matches_fixture(setup):
    case, srs, circuit, pp, cols = setup
    limbs = lambda vals: _lib.ints_to_limbs([int(v) % R for v in vals])
    sel = [limbs(col) for col in circuit.get_selector_polynomials()]
    sig = [limbs(col) for col in build_permutation_polynomials(pp.sigma, pp.n, pp.domain)]
    dev = DevicePlonk(sel, sig, g1_to_limbs(srs.g1_powers))
    dpp = dev.preprocessed()
    for k, v in case["preprocessed"].items():
        got = getattr(dpp, k)
        assert [str(int(got[0])), str(int(got[1]))] == v, k
    got = dev.prove(limbs(cols[0]), limbs(cols[1]), limbs(cols[2]), blinding=[int(v) for v in case["blinding"]])
    _assert_fields(got, case)
    ocirc, oa, ob, oc, opub, oblind, osrs, _ = load_case(case)
    assert pl.verify(_as_oracle_proof(got), opub, pl.preprocess(ocirc, osrs), osrs) is True


def _oracle_circuit_n64(which):
    """Two 64-gate circuits in the oracle's plain-int form (built with the reference's gate API, circuit.py:116-161): `chain` alternates
    multiplication and addition gates, each output wired to the next left input; `mixed` adds constant gates, a value fanned out to
    several right inputs and a tail of unwired gates."""
    import numpy as np
    rng = np.random.default_rng(64 if which == "chain" else 65)
    c = pl.Circuit()
    a, b, cc = [], [], []
    cur = int(rng.integers(2, 1 << 40))
    shared = int(rng.integers(2, 1 << 40))
    for i in range(64):
        kind = i % 2 if which == "chain" else (i * 7 + 3) % 3
        y = shared if (which == "mixed" and i % 5 == 0) else int(rng.integers(1, 1 << 40))
        if kind == 0:
            g, out = c.add_multiplication_gate(), cur * y % R
        elif kind == 1:
            g, out = c.add_addition_gate(), (cur + y) % R
        else:
            k = int(rng.integers(1, 1 << 30))
            g, out, y = c.add_constant_gate(k), (cur + k) % R, 0
        a.append(cur); b.append(y); cc.append(out)
        if i and not (which == "mixed" and i >= 56):
            c.add_copy_constraint(g - 1, 2, g, 0)              # previous output -> this left input
        if which == "mixed" and i % 5 == 0 and i and kind != 2:
            c.add_copy_constraint(0, 1, g, 1)                  # the shared right input
        cur = out
    assert pl.gates_satisfied(c, a, b, cc)
    return c, a, b, cc


@pytest.mark.parametrize("which,zero_row", [("chain", None), ("mixed", None), ("chain", 20), ("mixed", 63)])
def test_device_prover_equals_oracle_at_n64(which, zero_row):
    """DevicePlonk against oracle/plonk_ref.py (the reference's O(n^2) shape: coefficient products, long division by Z_H, one ec_mul per
    commitment term) at n = 64, all 16 proof fields and the 8 preprocessing commitments -- nothing here is compared with another
    part of the library.  zero_row: beta / gamma chosen so that this row of the grand product has a zero denominator
    (permutation.py:118-135 with py_ecc's x / 0 = 0; on the last row the reference never divides): the two provers must then
    behave alike -- the same proof, or both refuse because the constraint polynomial is no longer divisible by Z_H."""
    import c_oracle as co
    ocirc, a, b, c = _oracle_circuit_n64(which)
    n = 64
    osrs = o.srs_generate(n + 6, 7)
    opp = pl.preprocess(ocirc, osrs)
    assert opp.n == n
    blinding = [31337 + 101 * i for i in range(9)]
    challenges = None
    if zero_row is not None:
        s1 = pl.build_permutation_polynomials(opp.sigma, n, opp.domain)[0]
        beta = 0xB37A5EED
        challenges = {"beta": beta, "gamma": (-(a[zero_row] + beta * s1[zero_row])) % R}
    limbs = lambda vals: _lib.ints_to_limbs([int(v) % R for v in vals])
    sel = [limbs(col) for col in ocirc.get_selector_polynomials()]
    sig = [limbs(col) for col in pl.build_permutation_polynomials(opp.sigma, n, opp.domain)]
    dev = DevicePlonk(sel, sig, co.g1_to_arr(osrs[0]))
    dpp = dev.preprocessed()
    for k in ("q_l", "q_r", "q_o", "q_m", "q_c", "s_sigma1", "s_sigma2", "s_sigma3"):
        got = getattr(dpp, k + "_comm")
        assert (None if got is None else (int(got[0]), int(got[1]))) == getattr(opp, k + "_comm"), k
    try:
        want = pl.prove(ocirc, a, b, c, [], opp, osrs, blinding, challenges=challenges)
    except ValueError:
        want = None
    if want is None:
        assert zero_row is not None                                     # an ordinary proof must exist
        with pytest.raises(ValueError):
            dev.prove(limbs(a), limbs(b), limbs(c), blinding=blinding, challenges=challenges)
        return
    got = _as_oracle_proof(dev.prove(limbs(a), limbs(b), limbs(c), blinding=blinding, challenges=challenges))
    for f in pl.PROOF_FIELDS:
        assert getattr(got, f) == getattr(want, f), f
    if zero_row is None:
        assert pl.verify(got, [], opp, osrs) is True


def test_unsatisfied_witness_is_refused_like_the_reference(setup):
    """round3.py:140-147 raises when C is not divisible by Z_H; so do the oracle and both backends."""
    case, srs, circuit, pp, cols = setup
    bad_c = list(cols[2])
    bad_c[1] = bad_c[1] + FR(1)
    blinding = [int(v) for v in case["blinding"]]
    with pytest.raises(ValueError):
        prove(circuit, cols[0], cols[1], bad_c, [], pp, srs, blinding=blinding)
    ocirc, oa, ob, oc, opub, oblind, osrs, _ = load_case(case)
    oc[1] = (oc[1] + 1) % R
    with pytest.raises(ValueError):
        pl.prove(ocirc, oa, ob, oc, opub, pl.preprocess(ocirc, osrs), osrs, oblind)


def test_utils_and_polynomial_helpers_match_oracle():
    """The helpers the reference's PLONK modules import from utils.py / polynomial.py (utils.py:25-141,208-246,
    polynomial.py:189-261,438-475) on the backend types, against the oracle's plain-int forms."""
    from zkhip.field import get_root_of_unity, get_roots_of_unity
    from zkhip.plonk.polynomial import Polynomial, lagrange_basis, poly_div
    from zkhip.plonk.utils import (lagrange_basis_eval, next_power_of_2, pad_to_power_of_2, public_input_poly_eval, public_input_polynomial,
                                   vanishing_poly_eval)
    n = 8
    w = get_root_of_unity(n)
    dom = get_roots_of_unity(n)
    odom = o.get_roots_of_unity(n)
    for zeta in (5, 123456789, int(dom[3])):
        assert int(vanishing_poly_eval(n, FR(zeta))) == pl.vanishing_poly_eval(n, zeta)
        for i in (0, 3, 7):
            assert int(lagrange_basis_eval(i, n, w, zeta)) == pl.lagrange_basis_eval(i, n, odom[1], zeta)
    for i in (0, 5):
        assert [int(c) for c in lagrange_basis(dom, i).coeffs] == pl.lagrange_basis(odom, i)
    pub = [FR(35), FR(7), 11]
    pi = public_input_polynomial(pub, n, w)                                   # one inverse NTT on the GPU
    assert [int(c) for c in pi.coeffs] == pl.from_evaluations([35, 7, 11, 0, 0, 0, 0, 0], odom[1])
    assert public_input_poly_eval(pub, n, w, FR(99)) == pi.evaluate(FR(99))
    assert public_input_polynomial([], n, w) == Polynomial.zero()
    assert [next_power_of_2(k) for k in (0, 1, 2, 3, 4, 5, 8, 9)] == [1, 1, 2, 4, 4, 8, 8, 16]
    assert pad_to_power_of_2([FR(1), FR(2), FR(3)]) == [FR(1), FR(2), FR(3), FR(0)] and pad_to_power_of_2([1, 2], fill=9) == [1, 2]
    p = Polynomial([3, 0, 5, 0, 0, 7])
    assert len(p) == 6 and p.scale(FR(2)) == p * 2 and (1 - Polynomial([1, 1])) == Polynomial([0, -1]) and Polynomial([4]) == 4
    zh = Polynomial.vanishing(4)
    assert [int(c) for c in zh.coeffs] == pl.vanishing(4) and Polynomial.one() == Polynomial([1])
    assert (p * zh).divide_by_vanishing(4) == p
    with pytest.raises(ValueError):
        (p * zh + Polynomial([1])).divide_by_vanishing(4)
    q, r = poly_div(p, Polynomial([2, 1]))
    oq, orr = pl.poly_div([3, 0, 5, 0, 0, 7], [2, 1])
    assert [int(c) for c in q.coeffs] == oq and [int(c) for c in r.coeffs] == orr
