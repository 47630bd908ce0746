"""Conformance of zkhip.groth16 with what the reference's own test-suite expects: tests/groth16/test_poly_utils.py (host glue,
CPU), test_setup.py, test_proving.py, test_verifying.py and test_integration.py (backend, -m gpu) on the reference's fixture
(tests/groth16/conftest.py:39-158: the x^3 + x + 5 = 35 R1CS / QAP, toxic waste 3926 / 3604 / 2971 / 1357 / 3721, r = 4106,
s = 4565, public wires [0, 1]).  The R1CS front end and the float QAP interpolation are out of scope (their outputs are this
path's inputs): the QAP polynomials come from the committed fixture tests/golden/toy_groth16.json ("inputs")."""
import json
import os

import pytest

from zkhip.field import FQ, FR, G1, G2, ec_mul, ec_pairing
from zkhip.groth16.poly_utils import (_add_polys, _div_polys, _eval_poly, _multiply_polys, _multiply_vec_vec, _subtract_polys, ax_val, bx_val,
                                      cx_val, getFRPoly1D, getFRPoly2D, getNumGates, getNumWires, hx_val, hxr, zx_val)
from zkhip.groth16.proving import build_rpub_enum, proof_a, proof_b, proof_c
from zkhip.groth16.setup import sigma11, sigma12, sigma13, sigma14, sigma15, sigma21, sigma22
from zkhip.groth16.verifying import lhs, rhs, verify

PLACEHOLDER = (FQ(0), FQ(0))                                  # setup.py:39,50


# ------------------------------------------------------------------ host glue (CPU): test_poly_utils.py
def test_polynomial_helpers():
    assert _multiply_polys([FR(1), FR(1)], [FR(1), FR(1)])[:3] == [FR(1), FR(2), FR(1)]
    assert _add_polys([FR(1), FR(2)], [FR(3), FR(4)])[:2] == [FR(4), FR(6)]
    assert _subtract_polys([FR(5), FR(3)], [FR(1), FR(1)])[:2] == [FR(4), FR(2)]
    q, r = _div_polys([FR(-1), FR(0), FR(1)], [FR(-1), FR(1)])
    assert q[:2] == [FR(1), FR(1)]
    assert _eval_poly([FR(3), FR(2)], FR(4)) == FR(11)
    assert _multiply_vec_vec([FR(1), FR(2), FR(3)], [FR(4), FR(5), FR(6)]) == FR(32)
    assert _multiply_vec_vec([FR(0), FR(0)], [FR(5), FR(10)]) == FR(0)
    assert getNumWires([[1, 2], [3, 4], [5, 6]]) == 3 and getNumGates([[1, 2, 3], [4, 5, 6]]) == 3


def test_fr_conversion_rounds_floats():
    out = getFRPoly1D([1.0, 2.0, 3.0])
    assert all(isinstance(x, FR) for x in out) and out[0] == FR(1) and out[2] == FR(3)
    assert getFRPoly1D([1.7, 2.3]) == [FR(2), FR(2)]          # round(), not truncation (poly_utils.py getFRPoly1D)
    grid = getFRPoly2D([[1.0, 2.0], [3.0, 4.0]])
    assert grid[0][0] == FR(1) and grid[1][1] == FR(4)


def test_evaluation_helpers():
    poly2d = [[FR(1), FR(2), FR(3)], [FR(4), FR(5), FR(6)]]
    assert ax_val(poly2d, FR(2))[:2] == [FR(17), FR(38)]
    assert bx_val(poly2d, FR(2))[0] == FR(17) and cx_val(poly2d, FR(2))[0] == FR(17)
    assert zx_val([FR(1), FR(-1)], FR(3)) == FR(-2)
    assert hx_val([FR(2), FR(3)], FR(4)) == FR(14)


@pytest.fixture(scope="module")
def qap(golden_dir):
    with open(os.path.join(golden_dir, "toy_groth16.json")) as f:
        return json.load(f)["inputs"]


def test_hxr_divides_exactly_on_the_reference_qap(qap):
    Ax, Bx, Cx = getFRPoly2D(qap["Ap"]), getFRPoly2D(qap["Bp"]), getFRPoly2D(qap["Cp"])
    Hx, remainder = hxr(Ax, Bx, Cx, getFRPoly1D(qap["Z"]), qap["R"])
    assert all(v == FR(0) for v in remainder)
    # the identity of test_integration.py:55-70 at the toxic point: (R.A)(R.B) - R.C == Z * H
    x = FR(qap["x_val"])
    Rx = getFRPoly1D(qap["R"])
    lhs_val = _multiply_vec_vec(Rx, ax_val(Ax, x)) * _multiply_vec_vec(Rx, bx_val(Bx, x)) - _multiply_vec_vec(Rx, cx_val(Cx, x))
    assert lhs_val == zx_val(getFRPoly1D(qap["Z"]), x) * hx_val(Hx, x)
    # R1CS satisfaction of the witness the fixture carries is implied by the zero remainder; its public part is (1, 3) -> out 35
    assert qap["R"] == [1, 3, 35, 9, 27, 30]


def test_build_rpub_enum():
    r_vec = [FR(1), FR(3), FR(35), FR(9), FR(27), FR(30)]
    assert build_rpub_enum([0, 1], r_vec) == [(0, FR(1)), (1, FR(3))]
    assert build_rpub_enum([0, 2], [FR(10), FR(20), FR(30)]) == [(0, FR(10)), (2, FR(30))]


# ------------------------------------------------------------------ backend (GPU): setup, proving, verifying
@pytest.fixture(scope="module")
def pipe(qap):
    alpha, beta, gamma, delta, x_val = (FR(qap[k]) for k in ("alpha", "beta", "gamma", "delta", "x_val"))
    Ax, Bx, Cx = getFRPoly2D(qap["Ap"]), getFRPoly2D(qap["Bp"]), getFRPoly2D(qap["Cp"])
    Zx, Rx = getFRPoly1D(qap["Z"]), getFRPoly1D(qap["R"])
    Hx, remainder = hxr(Ax, Bx, Cx, Zx, qap["R"])
    numGates, numWires = getNumGates(Ax), getNumWires(Ax)
    Axv, Bxv, Cxv, Zxv = ax_val(Ax, x_val), bx_val(Bx, x_val), cx_val(Cx, x_val), zx_val(Zx, x_val)
    pub = qap["pub"]
    d = dict(alpha=alpha, beta=beta, gamma=gamma, delta=delta, numGates=numGates, numWires=numWires)
    d["s11"] = sigma11(alpha, beta, delta)
    d["s12"] = sigma12(numGates, x_val)
    d["s13"], d["VAL"] = sigma13(numWires, alpha, beta, gamma, Axv, Bxv, Cxv, pub_r_indexs=pub)
    d["s14"] = sigma14(numWires, alpha, beta, delta, Axv, Bxv, Cxv, pub_r_indexs=pub)
    d["s15"] = sigma15(numGates, delta, x_val, Zxv)
    d["s21"] = sigma21(beta, delta, gamma)
    d["s22"] = sigma22(numGates, x_val)
    r, s = FR(qap["r"]), FR(qap["s"])
    d["prf_A"] = proof_a(d["s11"], d["s12"], Ax, Rx, r)
    d["prf_B"] = proof_b(d["s21"], d["s22"], Bx, Rx, s)
    d["prf_C"] = proof_c(d["s11"], d["s12"], d["s14"], d["s15"], Bx, Rx, Hx, s, r, d["prf_A"], pub_r_indexs=pub)
    d["rx_pub"] = build_rpub_enum(pub, Rx)
    return d


@pytest.mark.gpu
def test_setup_elements(pipe):
    d = pipe
    assert d["s11"] == [ec_mul(G1, int(d["alpha"])), ec_mul(G1, int(d["beta"])), ec_mul(G1, int(d["delta"]))]
    assert len(d["s12"]) == d["numGates"] and d["s12"][0] == ec_mul(G1, 1)
    assert len(d["s13"]) == d["numWires"] and len(d["s14"]) == d["numWires"]
    for idx in (0, 1):                                        # public wires: real points in sigma1_3, placeholders in sigma1_4
        assert d["s13"][idx] is not None and d["s13"][idx] != PLACEHOLDER and d["s14"][idx] == PLACEHOLDER
    for idx in range(2, d["numWires"]):                       # private wires: the other way round
        assert d["s13"][idx] == PLACEHOLDER and d["s14"][idx] != PLACEHOLDER
    assert len(d["s15"]) == d["numGates"] - 1
    assert d["s21"] == [ec_mul(G2, int(d["beta"])), ec_mul(G2, int(d["gamma"])), ec_mul(G2, int(d["delta"]))]
    assert len(d["s22"]) == d["numGates"] and d["s22"][0] == ec_mul(G2, 1)


@pytest.mark.gpu
def test_proof_elements_are_points(pipe):
    for name in ("prf_A", "prf_B", "prf_C"):
        pt = pipe[name]
        assert pt is not None and isinstance(pt, tuple) and len(pt) == 2


@pytest.mark.gpu
def test_verifying(pipe):
    d = pipe
    assert lhs(d["prf_A"], d["prf_B"]) is not None and lhs(d["prf_A"], d["prf_B"]) == ec_pairing(d["prf_B"], d["prf_A"])
    assert rhs(d["prf_C"], d["s11"], d["s13"], d["s21"], d["rx_pub"]) is not None
    assert lhs(d["prf_A"], d["prf_B"]) == rhs(d["prf_C"], d["s11"], d["s13"], d["s21"], d["rx_pub"])
    assert verify(d["prf_A"], d["prf_B"], d["prf_C"], d["s11"], d["s13"], d["s21"], d["rx_pub"]) is True
    for fake_a in (ec_mul(G1, 9999), ec_mul(G1, 42)):         # test_verifying.py:42-50, test_integration.py:84-94
        assert verify(fake_a, d["prf_B"], d["prf_C"], d["s11"], d["s13"], d["s21"], d["rx_pub"]) is False
    assert verify(d["prf_A"], d["prf_B"], ec_mul(G1, 12345), d["s11"], d["s13"], d["s21"], d["rx_pub"]) is False
