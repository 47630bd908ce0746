"""CPU tests of the reference-shaped PLONK oracle (oracle/plonk_ref.py) against the committed fixture
tests/golden/plonk_proofs.json: the fixture is reproduced bit for bit, the oracle's verifier accepts it and rejects
each of the 16 single-field tamperings the reference's own suite applies (tests/plonk/test_e2e.py:198-254), the
polynomial primitives satisfy the identities the reference's tests state (tests/plonk/test_foundation.py), and the
transcript bytes are what zkp/plonk/transcript.py:36-123 prescribes."""
import copy
import hashlib
import json
import os

import pytest

import plonk_ref as pl
import py_ref as o

R = o.R


@pytest.fixture(scope="module")
def fixture(golden_dir):
    with open(os.path.join(golden_dir, "plonk_proofs.json")) as f:
        return json.load(f)["cases"]


def _pt(v):
    return None if v is None else (int(v[0]), int(v[1]))


def load_case(case):
    """fixture case -> (circuit, a, b, c, pub, blinding, srs, proof)."""
    c = pl.Circuit()
    c.gates = [tuple(int(v) for v in g) for g in case["gates"]]
    c.copy_constraints = [tuple(cc) for cc in case["copy_constraints"]]
    c.num_public_inputs = len(case["public_inputs"])
    ints = lambda key: [int(v) for v in case[key]]
    srs = o.srs_generate(case["srs"]["max_degree"], case["srs"]["seed"])
    proof = pl.Proof()
    for f, v in case["proof"].items():
        setattr(proof, f, _pt(v) if f in pl.PROOF_POINTS else int(v))
    return c, ints("a_vals"), ints("b_vals"), ints("c_vals"), ints("public_inputs"), ints("blinding"), srs, proof


@pytest.fixture(scope="module", params=["toy_x3", "six_gates_n8"])
def proved(request, fixture):
    case = fixture[request.param]
    circuit, a, b, c, pub, blinding, srs, want = load_case(case)
    pp = pl.preprocess(circuit, srs)
    proof, st = pl.prove(circuit, a, b, c, pub, pp, srs, blinding, return_state=True)
    return case, pp, srs, pub, proof, st, want


def test_fixture_is_reproduced(proved):
    case, pp, srs, pub, proof, st, want = proved
    assert pp.n == case["n"] and pp.sigma == case["sigma"]
    for k, v in case["preprocessed"].items():
        assert getattr(pp, k) == _pt(v), k
    for k, v in case["challenges"].items():
        assert getattr(st, k) == int(v), k
    for k, v in case["polys"].items():
        assert getattr(st, k) == [int(x) for x in v], k
    for f in pl.PROOF_FIELDS:
        assert getattr(proof, f) == getattr(want, f), f


def test_oracle_verifier_accepts_and_rejects_the_16_tamperings(proved):
    case, pp, srs, pub, proof, st, want = proved
    assert pl.verify(want, pub, pp, srs) is True
    if case["n"] != 4:
        return                                           # the tamper matrix on the reference's own circuit only (CPU-suite time)
    for f in pl.PROOF_SCALARS:                           # tests/plonk/test_e2e.py:205-222
        bad = copy.copy(want)
        setattr(bad, f, (getattr(want, f) + 1) % R)
        assert pl.verify(bad, pub, pp, srs) is False, f
    fake = o.ec_mul(o.G1, 0x1D0F4C0FFEE)
    for f in pl.PROOF_POINTS:                            # tests/plonk/test_e2e.py:234-254
        bad = copy.copy(want)
        setattr(bad, f, fake)
        assert pl.verify(bad, pub, pp, srs) is False, f


def test_openings_divide_exactly_and_split_matches_degrees(proved):
    case, pp, srs, pub, proof, st, want = proved
    n = pp.n
    assert st.rem_zeta == [0] and st.rem_zeta_omega == [0]
    assert len(st.a_poly) == n + 2 and len(st.z_poly) <= n + 3          # blinding degrees: round1.py:93-96, round2.py:74-79
    assert len(st.t_lo_poly) <= n and len(st.t_mid_poly) <= n and n < len(st.t_hi_poly) <= n + 6
    # z interpolates the accumulator and closes the product (permutation.py:126-137)
    zs = [pl.p_eval(st.z_poly, w) for w in pp.domain]
    assert zs == pl.compute_accumulator(st.a_vals, st.b_vals, st.c_vals, pp.sigma, n, pp.domain, st.beta, st.gamma)
    assert zs[0] == 1


def test_polynomial_primitives():
    a, b = pl.P([1, 2, 3]), pl.P([5, 0, 0, 7])
    prod = pl.p_mul(a, b)
    for x in (0, 1, 2, 12345, R - 3):
        assert pl.p_eval(prod, x) == pl.p_eval(a, x) * pl.p_eval(b, x) % R
        assert pl.p_eval(pl.p_add(a, b), x) == (pl.p_eval(a, x) + pl.p_eval(b, x)) % R
        assert pl.p_eval(pl.p_sub(a, b), x) == (pl.p_eval(a, x) - pl.p_eval(b, x)) % R
    q, r = pl.poly_div(pl.p_add(prod, pl.P([4, 1])), b)
    assert q == a and r == pl.P([4, 1])
    assert pl.poly_div(pl.P([R - 1, 0, 1]), pl.P([R - 1, 1])) == (pl.P([1, 1]), pl.P([0]))   # (x^2-1)/(x-1): polynomial.py:404-408
    assert pl.poly_div(a, b) == (pl.P([0]), a)
    with pytest.raises(ValueError):
        pl.poly_div(a, pl.P([0]))
    assert pl.P([3, 0, 0]) == [3] and pl.P([]) == [0] and pl.P([R, R + 1]) == [0, 1]
    dom = o.get_roots_of_unity(8)
    for i in (0, 3, 7):
        li = pl.lagrange_basis(dom, i)
        assert [pl.p_eval(li, d) for d in dom] == [1 if j == i else 0 for j in range(8)]
        for zeta in (5, 987654321):
            assert pl.lagrange_basis_eval(i, 8, dom[1], zeta) == pl.p_eval(li, zeta)
        assert pl.lagrange_basis_eval(i, 8, dom[1], dom[i]) == 1
    zh = pl.vanishing(8)
    assert all(pl.p_eval(zh, d) == 0 for d in dom) and pl.p_eval(zh, 3) == pl.vanishing_poly_eval(8, 3)
    evals = [7, 1, 0, R - 1, 5, 5, 9, 2]
    poly = pl.from_evaluations(evals, dom[1])
    assert [pl.p_eval(poly, d) for d in dom] == evals


def test_transcript_bytes():
    t = pl.Transcript()
    t.append_point(b"a_comm", o.G1)
    t.append_scalar(b"a_eval", R + 5)
    t.append_point(b"z", None)
    want = b"plonk" + b"a_comm" + (1).to_bytes(32, "big") + (2).to_bytes(32, "big") + b"a_eval" + (5).to_bytes(32, "big") + b"z" + bytes(64)
    assert bytes(t.state) == want
    ch = t.challenge_scalar(b"beta")
    dig = hashlib.sha256(want + b"beta").digest()
    assert ch == int.from_bytes(dig, "big") % R and bytes(t.state) == want + b"beta" + dig
    assert t.challenge_scalar(b"gamma") == int.from_bytes(hashlib.sha256(want + b"beta" + dig + b"gamma").digest(), "big") % R


def test_second_circuit_is_consistent():
    circuit, a, b, c, pub = pl.circuit_six_gates()
    assert pl.gates_satisfied(circuit, a, b, c)
    pl.preprocess(circuit, o.srs_generate(8, 1))                                          # pads to 8 gates in place
    assert circuit.n == 8 and pl.gates_satisfied(circuit, a, b, c)
    sigma = circuit.build_copy_constraints()
    assert sorted(sigma) == list(range(24))
    vals = a + b + c
    assert all(vals[i] == vals[sigma[i]] for i in range(24))
    assert sum(1 for i, s_ in enumerate(sigma) if s_ != i) == 16                          # the 16 wire positions tied by the 10 constraints
