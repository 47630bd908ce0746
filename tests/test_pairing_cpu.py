"""CPU tests of the host-side pairing in libzkhip (csrc/pairing.hip; verifier support, SURVEY.md
section 8 f1) against the oracle's restatement of py_ecc's algorithm (oracle/py_ref.py: FQ12 polynomial
arithmetic, twist, linefunc, Miller loop, final exponentiation) -- coefficient by coefficient."""
import ctypes

import numpy as np

import c_oracle as co
import py_ref as o
from zkhip import _lib


def pairing(P, Q):
    out = np.zeros(48, dtype=np.uint64)
    rc = _lib.load().zk_pairing(_lib.ptr(co.g1_to_arr([P])), _lib.ptr(co.g2_to_arr([Q])), _lib.ptr(out))
    assert rc == 0
    return co.from_limbs(out)


def check(pairs):
    ok = ctypes.c_int(-1)
    g1 = co.g1_to_arr([p for p, _ in pairs])
    g2 = co.g2_to_arr([q for _, q in pairs])
    assert _lib.load().zk_pairing_check(_lib.ptr(g1), _lib.ptr(g2), len(pairs), ctypes.byref(ok)) == 0
    return bool(ok.value)


def test_pairing_matches_oracle_coefficients():
    e = pairing(o.G1, o.G2)
    assert e == o.pairing(o.G2, o.G1)
    assert e != o.F12_ONE
    P5, Q7 = o.g1_multiply(o.G1, 5), o.g2_multiply(o.G2, 7)
    assert pairing(P5, Q7) == o.pairing(Q7, P5)
    big = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % o.R
    Pb, Qb = co.g1_mul(o.G1, big), co.g2_mul(o.G2, big + 1)
    assert pairing(Pb, Qb) == o.pairing(Qb, Pb)


def test_bilinearity_and_identity():
    e = pairing(o.G1, o.G2)
    assert pairing(o.g1_multiply(o.G1, 2), o.G2) == o.f12_mul(e, e)
    assert pairing(o.G1, o.g2_multiply(o.G2, 2)) == o.f12_mul(e, e)
    assert o.f12_pow(e, o.R) == o.F12_ONE
    assert pairing(None, o.G2) == o.F12_ONE and pairing(o.G1, None) == o.F12_ONE   # infinity inputs


def test_pairing_check_products():
    P5, Q7 = o.g1_multiply(o.G1, 5), o.g2_multiply(o.G2, 7)
    assert check([(P5, Q7), (o.g1_neg(o.g1_multiply(o.G1, 35)), o.G2)])
    assert not check([(P5, Q7), (o.g1_neg(o.g1_multiply(o.G1, 36)), o.G2)])
    assert check([]) and check([(None, o.G2)])


def test_groth16_verify_equation_on_the_toy_proof():
    """zkp/groth16/verifying.py:29-40 on the toy circuit: the oracle verifies it, and the library's
    product form agrees; a tampered proof element is rejected by both."""
    d = o.toy_groth16()
    rx_pub = [(i, d["Rx"][i]) for i in o.TOY["pub"]]
    temp = None
    for i, ri in rx_pub:
        temp = o.g1_add(temp, o.g1_multiply(d["s13"][i], ri))
    pairs = [(d["proof_A"], d["proof_B"]), (o.g1_neg(d["s11"][0]), d["s21"][0]), (o.g1_neg(temp), d["s21"][1]),
             (o.g1_neg(d["proof_C"]), d["s21"][2])]
    assert check(pairs)
    assert o.groth16_verify(d["proof_A"], d["proof_B"], d["proof_C"], d["s11"], d["s13"], d["s21"], rx_pub)
    bad = [(o.g1_double(d["proof_A"]), d["proof_B"])] + pairs[1:]
    assert not check(bad)


def test_off_curve_and_non_canonical_inputs_are_rejected():
    """py_ecc's pairing asserts is_on_curve(Q, b2) and is_on_curve(P, b) (bn128_pairing.py); the library refuses the same
    inputs with ZK_ERR_INVALID (the Python facade turns that into the reference's AssertionError), and the oracle's
    restatement carries the assertions too.  The all-zero encoding stays the point at infinity."""
    import pytest
    from zkhip.field import FQ, FQ2, ec_pairing, pairing_check
    lib = _lib.load()
    out = np.zeros(48, dtype=np.uint64)
    ok = ctypes.c_int(-1)
    g1, g2 = co.g1_to_arr([o.G1]), co.g2_to_arr([o.G2])
    bad_g1 = g1.copy()
    bad_g1[0, 4] += 1                                       # (1, 3): y^2 != x^3 + 3
    bad_g2 = g2.copy()
    bad_g2[0, 8] ^= 1                                       # y.c0 with its low bit flipped
    big_g1 = g1.copy()
    big_g1[0, :4] = co.to_limbs([o.P + 1])[0]               # x = p + 1: the residue is on the curve, the encoding is not canonical
    for p_arr, q_arr in ((bad_g1, g2), (g1, bad_g2), (big_g1, g2)):
        assert lib.zk_pairing(_lib.ptr(p_arr), _lib.ptr(q_arr), _lib.ptr(out)) == _lib.ZK_ERR_INVALID
        assert b"canonical point" in lib.zk_last_error()
        assert lib.zk_pairing_check(_lib.ptr(p_arr), _lib.ptr(q_arr), 1, ctypes.byref(ok)) == _lib.ZK_ERR_INVALID
    # a valid pair in front does not let a bad one through
    two_p, two_q = np.concatenate([g1, bad_g1]), np.concatenate([g2, g2])
    assert lib.zk_pairing_check(_lib.ptr(two_p), _lib.ptr(two_q), 2, ctypes.byref(ok)) == _lib.ZK_ERR_INVALID
    # facade: the reference's AssertionError
    off1 = (FQ(1), FQ(3))
    off2 = (FQ2([o.G2[0][0], o.G2[0][1]]), FQ2([o.G2[1][0] ^ 1, o.G2[1][1]]))
    good2 = (FQ2(list(o.G2[0])), FQ2(list(o.G2[1])))
    with pytest.raises(AssertionError):
        ec_pairing(good2, off1)
    with pytest.raises(AssertionError):
        ec_pairing(off2, (FQ(1), FQ(2)))
    with pytest.raises(AssertionError):
        pairing_check([((FQ(1), FQ(2)), good2), (off1, good2)])
    assert ec_pairing(good2, None) == ec_pairing(None, (FQ(1), FQ(2)))    # infinity still gives the identity
    # the oracle mirrors the assertion
    with pytest.raises(AssertionError):
        o.pairing(o.G2, (1, 3))
    with pytest.raises(AssertionError):
        o.pairing(((o.G2[0][0], o.G2[0][1]), (o.G2[1][0] ^ 1, o.G2[1][1])), o.G1)
