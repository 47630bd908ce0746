"""GPU tests (-m gpu) of the Python facade that mirrors the reference's call signatures
(zkp.groth16.setup / proving / poly_utils, zkp.plonk.field / polynomial / utils / kzg / srs).
The cases restate the reference's own tests (tests/groth16/test_setup.py, test_proving.py,
test_integration.py; tests/plonk/test_foundation.py:486-540,724-760; tests/plonk/test_crypto.py:41-191)
and compare against the committed golden fixtures, so they read like the reference's suite."""
import json
import os

import pytest

import py_ref as o
from zkhip.field import (FQ, FQ2, FQ12, FR, G1, G2, Z1, CURVE_ORDER, ec_add, ec_mul, ec_neg, ec_pairing, get_root_of_unity,
                         get_roots_of_unity)
from zkhip.groth16.poly_utils import (ax_val, bx_val, cx_val, getFRPoly1D, getFRPoly2D, getNumGates, getNumWires,
                                      hx_val, hxr, zx_val)
from zkhip.groth16.proving import build_rpub_enum, proof_a, proof_b, proof_c
from zkhip.groth16.setup import sigma11, sigma12, sigma13, sigma14, sigma15, sigma21, sigma22
from zkhip.groth16.verifying import lhs, rhs, verify
from zkhip.plonk.kzg import commit, create_witness, verify_opening
from zkhip.plonk.polynomial import Polynomial, fft, ifft, poly_div
from zkhip.plonk.srs import SRS
from zkhip.plonk.utils import coset_fft, coset_ifft

pytestmark = pytest.mark.gpu


def g1j(v):
    return None if v is None else (FQ(int(v[0])), FQ(int(v[1])))


def g2j(v):
    return None if v is None else (FQ2((int(v[0][0]), int(v[0][1]))), FQ2((int(v[1][0]), int(v[1][1]))))


@pytest.fixture(scope="module")
def toy(golden_dir):
    return json.load(open(os.path.join(golden_dir, "toy_groth16.json")))


@pytest.fixture(scope="module")
def kzg_golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "kzg_seed42.json")))


# ------------------------------------------------------------------ Groth16 (tests/groth16/conftest.py:82-158)
@pytest.fixture(scope="module")
def pipeline(toy):
    t = toy["inputs"]
    alpha, beta, gamma, delta, x_val = (FR(t[k]) for k in ("alpha", "beta", "gamma", "delta", "x_val"))
    Ax, Bx, Cx = getFRPoly2D(t["Ap"]), getFRPoly2D(t["Bp"]), getFRPoly2D(t["Cp"])
    Zx, Rx = getFRPoly1D(t["Z"]), getFRPoly1D(t["R"])
    Hx, remainder = hxr(Ax, Bx, Cx, Zx, t["R"])
    numGates, numWires = getNumGates(Ax), getNumWires(Ax)
    Axv, Bxv, Cxv, Zxv = ax_val(Ax, x_val), bx_val(Bx, x_val), cx_val(Cx, x_val), zx_val(Zx, x_val)
    s11 = sigma11(alpha, beta, delta)
    s12 = sigma12(numGates, x_val)
    s13, VAL = sigma13(numWires, alpha, beta, gamma, Axv, Bxv, Cxv, pub_r_indexs=t["pub"])
    s14 = sigma14(numWires, alpha, beta, delta, Axv, Bxv, Cxv, pub_r_indexs=t["pub"])
    s15 = sigma15(numGates, delta, x_val, Zxv)
    s21 = sigma21(beta, delta, gamma)
    s22 = sigma22(numGates, x_val)
    r, s = FR(t["r"]), FR(t["s"])
    prf_A = proof_a(s11, s12, Ax, Rx, r)
    prf_B = proof_b(s21, s22, Bx, Rx, s)
    prf_C = proof_c(s11, s12, s14, s15, Bx, Rx, Hx, s, r, prf_A, pub_r_indexs=t["pub"])
    return dict(locals())


def test_hxr_matches_golden(pipeline, toy):
    assert [str(int(v)) for v in pipeline["Hx"]] == toy["Hx"]
    assert all(int(v) == 0 for v in pipeline["remainder"])           # tests/groth16/test_poly_utils.py
    assert len(pipeline["Hx"]) == 2 * 6 - 1 - 4                      # quirk: length 2W-1-G
    assert str(int(hx_val(pipeline["Hx"], pipeline["x_val"]))) == str(o.eval_poly([int(v) for v in toy["Hx"]], 3721))


def test_setup_sigmas_match_golden(pipeline, toy):
    assert pipeline["s11"] == [g1j(p) for p in toy["sigma1_1"]]
    assert pipeline["s12"] == [g1j(p) for p in toy["sigma1_2"]]
    assert pipeline["s13"] == [g1j(p) for p in toy["sigma1_3"]]
    assert pipeline["s14"] == [g1j(p) for p in toy["sigma1_4"]]
    assert pipeline["s15"] == [g1j(p) for p in toy["sigma1_5"]]
    assert pipeline["s21"] == [g2j(p) for p in toy["sigma2_1"]]
    assert pipeline["s22"] == [g2j(p) for p in toy["sigma2_2"]]
    assert [str(int(v)) for v in pipeline["VAL"]] == toy["VAL"]


def test_setup_relations(pipeline):
    """tests/groth16/test_setup.py:15-28,37-40,96-121."""
    p = pipeline
    assert p["s11"][0] == ec_mul(G1, int(p["alpha"])) and p["s11"][2] == ec_mul(G1, int(p["delta"]))
    assert p["s12"][0] == G1 and len(p["s12"]) == p["numGates"]
    assert p["s13"][2] == (FQ(0), FQ(0)) and p["s14"][0] == (FQ(0), FQ(0))     # placeholders
    assert len(p["s15"]) == p["numGates"] - 1 and len(p["s22"]) == p["numGates"]
    assert p["s21"][1] == ec_mul(G2, int(p["gamma"]))


def test_proof_elements_match_golden(pipeline, toy):
    for pt in (pipeline["prf_A"], pipeline["prf_B"], pipeline["prf_C"]):
        assert isinstance(pt, tuple) and len(pt) == 2                            # tests/groth16/test_proving.py:13-46
    assert pipeline["prf_A"] == g1j(toy["proof_A"])
    assert pipeline["prf_B"] == g2j(toy["proof_B"])
    assert pipeline["prf_C"] == g1j(toy["proof_C"])
    # closed form of zkp/groth16/test.py:303-325
    assert pipeline["prf_A"] == ec_mul(G1, int(toy["A"]))
    assert pipeline["prf_B"] == ec_mul(G2, int(toy["B"]))
    assert pipeline["prf_C"] == ec_mul(G1, int(toy["C"]))
    assert build_rpub_enum([0, 1], pipeline["Rx"]) == [(0, FR(1)), (1, FR(3))]


def test_proof_c_other_public_indices(pipeline, toy):
    """zkp/groth16/arb_private: pub = [0, 5] (C given in SURVEY.md appendix B)."""
    p, t = pipeline, toy["inputs"]
    s14 = sigma14(p["numWires"], p["alpha"], p["beta"], p["delta"], p["Axv"], p["Bxv"], p["Cxv"], pub_r_indexs=[0, 5])
    c = proof_c(p["s11"], p["s12"], s14, p["s15"], p["Bx"], p["Rx"], p["Hx"], p["s"], p["r"], p["prf_A"], pub_r_indexs=[0, 5])
    assert c == ec_mul(G1, 1822676082916608769428035261027319148862170359061910007020342316389334981081)


# ------------------------------------------------------------------ EC wrappers (tests/plonk/test_foundation.py)
def test_ec_wrappers(kzg_golden):
    two_g1 = g1j(kzg_golden["two_G1"])
    assert ec_add(G1, G1) == two_g1 == ec_mul(G1, 2) == ec_mul(G1, FR(2))
    assert ec_mul(G1, 0) is None and ec_mul(G1, CURVE_ORDER) is None and ec_mul(None, 5) is None
    assert ec_add(G1, None) == G1 and ec_add(None, G1) == G1 and ec_add(G1, ec_neg(G1)) is None
    assert ec_mul(G1, CURVE_ORDER + 3) == ec_mul(G1, 3)                         # reduced mod r (field.py:86-88)
    assert ec_add(ec_mul(G2, 3), ec_mul(G2, 4)) == ec_mul(G2, 7)
    assert ec_neg(None) is None and Z1 is None


def test_roots_of_unity():
    w = get_root_of_unity(4)
    assert w ** 4 == FR(1) and w ** 2 != FR(1)
    roots = get_roots_of_unity(8)
    assert len(roots) == 8 and roots[0] == FR(1) and all(r ** 8 == FR(1) for r in roots)
    assert get_root_of_unity(1) == FR(1)
    with pytest.raises(ValueError):
        get_root_of_unity(6)
    with pytest.raises(ValueError):
        get_root_of_unity(1 << 29)


# ------------------------------------------------------------------ FFT (tests/plonk/test_foundation.py:486-540)
def test_fft_single():
    assert fft([FR(7)], FR(1)) == [FR(7)]
    assert ifft([FR(7)], FR(1)) == [FR(7)]


def test_fft_basic_and_all_points():
    n = 4
    omega = get_root_of_unity(n)
    coeffs = [FR(1), FR(2), FR(3), FR(4)]
    evals = fft(coeffs, omega)
    assert len(evals) == n and evals[0] == Polynomial(coeffs).evaluate(FR(1)) == FR(10)
    n = 8
    omega = get_root_of_unity(n)
    coeffs = [FR(i) for i in range(n)]
    evals = fft(coeffs, omega)
    p = Polynomial(coeffs)
    for i in range(n):
        assert evals[i] == p.evaluate(omega ** i)


def test_fft_ifft_roundtrips():
    n = 8
    omega = get_root_of_unity(n)
    original = [FR(i * 3 + 1) for i in range(n)]
    assert ifft(fft(original, omega), omega) == original
    original = [FR(7), FR(11), FR(13), FR(17)]
    omega = get_root_of_unity(4)
    assert fft(ifft(original, omega), omega) == original
    evals = [FR(10), FR(5), FR(3), FR(7)]
    assert fft(ifft(evals, omega), omega) == evals


def test_fft_other_primitive_root():
    """Callers may pass any primitive n-th root (e.g. omega^-1 or omega^3)."""
    n = 8
    omega = get_root_of_unity(n)
    coeffs = [FR(5 * i + 2) for i in range(n)]
    for e in (3, 5, 7):
        w = omega ** e
        ev = fft(coeffs, w)
        p = Polynomial(coeffs)
        assert all(ev[i] == p.evaluate(w ** i) for i in range(n))
        assert ifft(ev, w) == coeffs
    with pytest.raises(ValueError):
        fft(coeffs, omega * omega)  # not primitive


def test_coset_fft(golden_dir):
    n = 8
    omega = get_root_of_unity(n)
    coeffs = [FR(i + 1) for i in range(n)]
    ev = coset_fft(coeffs, omega)
    p = Polynomial(coeffs)
    assert all(ev[i] == p.evaluate(FR(5) * omega ** i) for i in range(n))       # tests/plonk/test_foundation.py:724-760
    assert coset_ifft(ev, omega) == coeffs
    ev7 = coset_fft(coeffs, omega, FR(7))
    assert all(ev7[i] == p.evaluate(FR(7) * omega ** i) for i in range(n))
    assert coset_ifft(ev7, omega, FR(7)) == coeffs


def test_from_evaluations_and_poly_div():
    omega = get_root_of_unity(4)
    p = Polynomial([FR(1), FR(2)])
    evals = [p.evaluate(omega ** i) for i in range(4)]
    assert Polynomial.from_evaluations(evals, omega) == p                       # trims trailing zeros
    a = Polynomial([FR(-1), FR(0), FR(1)])
    q, r = poly_div(a, Polynomial([FR(-1), FR(1)]))
    assert q == Polynomial([FR(1), FR(1)]) and r.is_zero()


# ------------------------------------------------------------------ SRS / KZG (tests/plonk/test_crypto.py:41-191)
@pytest.fixture(scope="module")
def srs_small():
    return SRS.generate(max_degree=8, seed=42)


def test_srs_generate(srs_small, kzg_golden):
    assert len(srs_small.g1_powers) == 9 and srs_small.g1_powers[0] == G1 and srs_small.max_degree == 8
    assert srs_small.g1_powers == [g1j(p) for p in kzg_golden["g1_powers"]]
    assert srs_small.g2_powers == [g2j(p) for p in kzg_golden["g2_powers"]]
    tau = int(kzg_golden["tau"])
    assert srs_small.g1_powers[1] == ec_mul(G1, tau) and srs_small.g2_powers[1] == ec_mul(G2, tau)
    s0 = SRS.generate(max_degree=0, seed=42)
    assert len(s0.g1_powers) == 1 and s0.g1_powers[0] == G1


def test_device_srs_equals_the_golden_string_and_the_oracle(kzg_golden):
    """zkhip.plonk.srs.DeviceSRS: the same [tau^i]_1 (zkp/plonk/srs.py:68-85) with the exponents and the points produced on the
    device -- the golden string of seed 42 at the reference's size, and at 5000 powers (the table kernel of the fixed-base batch)
    against the oracle's tau^i * G1 and the host-buffer batch."""
    import numpy as np
    import c_oracle as co
    from zkhip import _lib
    from zkhip.field import g1_to_limbs
    from zkhip.plonk.srs import DeviceSRS
    d = DeviceSRS.generate(max_degree=8, seed=42).to_host()
    assert d.g1_powers == [g1j(p) for p in kzg_golden["g1_powers"]] and d.g2_powers == [g2j(p) for p in kzg_golden["g2_powers"]]
    tau = int(kzg_golden["tau"])
    big = DeviceSRS.generate(max_degree=4999, tau=tau, keep_tau=True)
    assert big.tau == tau and DeviceSRS.generate(max_degree=3, seed=42).tau is None       # the toxic value is not kept by default
    pts = big.d_g1.cpu().numpy().view(np.uint64)
    idx = [0, 1, 2, 2500, 4999]
    assert np.array_equal(pts[idx], co.g1_fixed_base_arr(o.G1, co.to_limbs([pow(tau, i, o.R) for i in idx])))
    powers = _lib.ints_to_limbs([pow(tau, i, o.R) for i in range(5000)])
    host = np.zeros((5000, 8), dtype=np.uint64)
    _lib.check(_lib.load().zk_fixed_base_g1(_lib.ptr(g1_to_limbs([G1])), _lib.ptr(powers), 5000, _lib.ptr(host)))
    assert np.array_equal(pts, host)


def test_commit_cases(srs_small, kzg_golden):
    for name, c in kzg_golden["commits"].items():
        poly = Polynomial([FR(int(v)) for v in c["coeffs"]])
        assert commit(poly, srs_small) == g1j(c["commitment"]), name
    assert commit(Polynomial([FR(7)]), srs_small) == ec_mul(G1, FR(7))
    a, b = FR(3), FR(5)
    assert commit(Polynomial([a, b]), srs_small) == ec_add(ec_mul(srs_small.g1_powers[0], a), ec_mul(srs_small.g1_powers[1], b))
    C = commit(Polynomial([FR(0)]), srs_small)
    assert C is None or C == Z1
    with pytest.raises(ValueError):
        commit(Polynomial([FR(1)] * 10), srs_small)
    assert commit(Polynomial([FR(0)] * 8 + [FR(1)]), srs_small) is not None


def test_commit_linearity(srs_small):
    p, q = Polynomial([FR(1), FR(2)]), Polynomial([FR(3), FR(4)])
    assert commit(p + q, srs_small) == ec_add(commit(p, srs_small), commit(q, srs_small))
    p, q = Polynomial([FR(1)]), Polynomial([FR(0), FR(0), FR(5)])
    assert commit(p + q, srs_small) == ec_add(commit(p, srs_small), commit(q, srs_small))
    p = Polynomial([FR(3), FR(7)])
    assert commit(p + Polynomial([FR(0)]), srs_small) == commit(p, srs_small)
    poly, s = Polynomial([FR(2), FR(3)]), FR(5)
    assert commit(poly * s, srs_small) == ec_mul(commit(poly, srs_small), s)


# ------------------------------------------------------------------ verify (tests/groth16/test_verifying.py, test_integration.py:59-79)
def test_groth16_verify_true_and_tampered(pipeline):
    p = pipeline
    rx_pub = build_rpub_enum(p["t"]["pub"], p["Rx"])
    assert verify(p["prf_A"], p["prf_B"], p["prf_C"], p["s11"], p["s13"], p["s21"], rx_pub) is True
    assert lhs(p["prf_A"], p["prf_B"]) == rhs(p["prf_C"], p["s11"], p["s13"], p["s21"], rx_pub)
    assert verify(ec_add(p["prf_A"], G1), p["prf_B"], p["prf_C"], p["s11"], p["s13"], p["s21"], rx_pub) is False
    assert verify(p["prf_A"], ec_add(p["prf_B"], G2), p["prf_C"], p["s11"], p["s13"], p["s21"], rx_pub) is False
    assert verify(p["prf_A"], p["prf_B"], ec_mul(p["prf_C"], 2), p["s11"], p["s13"], p["s21"], rx_pub) is False
    wrong_pub = [(0, FR(1)), (1, FR(4))]                                       # wrong public input
    assert verify(p["prf_A"], p["prf_B"], p["prf_C"], p["s11"], p["s13"], p["s21"], wrong_pub) is False


def test_public_curve_constants_through_the_backend():
    """The external anchors of tests/test_oracle.py::test_public_curve_constants, computed by the GPU backend: 2*G1 of
    alt_bn128 (EIP-196 vectors / py_ecc.bn128.double(G1)) by addition, scalar multiplication, a two-term MSM and the
    fixed-base batch; the published 2^28-th root of unity behind every NTT domain."""
    from zkhip.field import fixed_base_mul, msm_g1
    two_g = (FQ(1368015179489954701390400359078579693043519447331113978918064868415326638035),
             FQ(9918110051302171585080402603319702774565515993150576347155970296011118125764))
    assert ec_add(G1, G1) == two_g and ec_mul(G1, 2) == two_g
    assert msm_g1([FR(1), FR(1)], [G1, G1]) == two_g and fixed_base_mul(G1, [2, CURVE_ORDER + 2])[0] == two_g
    w28 = 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert int(get_root_of_unity(1 << 28)) == w28
    n = 16
    ev = fft([FR(0), FR(1)] + [FR(0)] * (n - 2), get_root_of_unity(n))       # p(x) = x evaluated on the domain: the powers of omega_16
    assert [int(v) for v in ev] == [pow(w28, (1 << 24) * i, CURVE_ORDER) for i in range(n)]


def test_backend_against_sympy_directly():
    """The GPU backend against a third-party implementation with no oracle in between: G1 scalar multiplications, an MSM and a
    commitment against SymPy's EllipticCurve over GF(p); fft / ifft against sympy.discrete.transforms.ntt / intt (primitive
    root 5, natural order -- the reference's convention)."""
    import random
    sympy = pytest.importorskip("sympy")
    from sympy.discrete.transforms import intt as sym_intt, ntt as sym_ntt
    from sympy.ntheory.elliptic_curve import EllipticCurve
    from zkhip.field import FIELD_MODULUS, msm_g1
    curve = EllipticCurve(0, 3, modulus=FIELD_MODULUS)
    gen = curve(1, 2)
    sym = lambda k: (lambda q: (FQ(int(q.x)), FQ(int(q.y))))((k % CURVE_ORDER) * gen)
    rng = random.Random(41)
    ks = [rng.randrange(CURVE_ORDER) for _ in range(6)]
    pts = [ec_mul(G1, k) for k in ks]
    assert pts == [sym(k) for k in ks]
    ss = [rng.randrange(CURVE_ORDER) for _ in range(6)]
    assert msm_g1([FR(v) for v in ss], pts) == sym(sum(a * b for a, b in zip(ss, ks)))
    srs = SRS.generate(max_degree=5, seed=42)
    tau = int.from_bytes(__import__("hashlib").sha256(b"42").digest(), "big") % CURVE_ORDER          # zkp/plonk/srs.py:68-70
    coeffs = [rng.randrange(CURVE_ORDER) for _ in range(6)]
    assert commit(Polynomial(coeffs), srs) == sym(sum(c * pow(tau, j, CURVE_ORDER) for j, c in enumerate(coeffs)))
    for n in (8, 64, 1024):
        seq = [rng.randrange(CURVE_ORDER) for _ in range(n)]
        w = get_root_of_unity(n)
        want = [int(v) for v in sym_ntt(seq, prime=CURVE_ORDER)]
        assert [int(v) for v in fft(seq, w)] == want
        assert [int(v) for v in ifft(want, w)] == seq == [int(v) for v in sym_intt(want, prime=CURVE_ORDER)]


def test_verifiers_refuse_off_curve_proof_points(pipeline, srs_small):
    """py_ecc's pairing asserts is_on_curve for both arguments, so the reference's verifiers raise on a proof element that
    is not a curve point; the backend refuses the same inputs (AssertionError from the facade, ZK_ERR_INVALID at the ABI)
    instead of running them through the Miller loop."""
    p = pipeline
    rx_pub = build_rpub_enum(p["t"]["pub"], p["Rx"])
    off_a = (p["prf_A"][0], p["prf_A"][1] + FQ(1))
    off_c = (p["prf_C"][0] + FQ(1), p["prf_C"][1])
    off_b = (p["prf_B"][0], FQ2((p["prf_B"][1].coeffs[0] + FQ(1), p["prf_B"][1].coeffs[1])))
    for args in ((off_a, p["prf_B"], p["prf_C"]), (p["prf_A"], off_b, p["prf_C"]), (p["prf_A"], p["prf_B"], off_c)):
        with pytest.raises(AssertionError):
            verify(args[0], args[1], args[2], p["s11"], p["s13"], p["s21"], rx_pub)
    with pytest.raises(AssertionError):
        lhs(off_a, p["prf_B"])
    poly = Polynomial([FR(1), FR(2), FR(3)])
    C, pi = commit(poly, srs_small), create_witness(poly, FR(5), srs_small)
    with pytest.raises(AssertionError):
        verify_opening(C, (pi[0], pi[1] + FQ(1)), FR(5), poly.evaluate(FR(5)), srs_small)


def test_pairing_wrapper_bilinear():
    e = ec_pairing(G2, G1)
    assert isinstance(e, FQ12) and e != FQ12.one()
    assert ec_pairing(G2, ec_mul(G1, 5)) == e ** 5 == ec_pairing(ec_mul(G2, 5), G1)
    assert ec_pairing(ec_mul(G2, 3), ec_mul(G1, 4)) == e ** 12
    assert e ** CURVE_ORDER == FQ12.one()


# ------------------------------------------------------------------ KZG openings (tests/plonk/test_crypto.py:198-301)
def test_kzg_openings(srs_small):
    p = Polynomial([FR(1), FR(2), FR(3), FR(4)])
    C = commit(p, srs_small)
    for z in (FR(0), FR(3), FR(7), FR(123456789)):
        pi = create_witness(p, z, srs_small)
        y = p.evaluate(z)
        assert verify_opening(C, pi, z, y, srs_small) is True
        assert verify_opening(C, pi, z, y + FR(1), srs_small) is False          # wrong evaluation
        assert verify_opening(C, pi, z + FR(1), y, srs_small) is False          # wrong point
    const = Polynomial([FR(9)])
    assert verify_opening(commit(const, srs_small), create_witness(const, FR(5), srs_small), FR(5), FR(9), srs_small)


def test_host_buffer_entry_points_reuse_their_plans(srs_small):
    """fft / ifft / commit go through zk_ntt_fr / zk_msm_g1 once per call: the NTT tables of a size and the MSM workspace of a size
    class are built by the FIRST call only (zk_cache_stats), a different size builds its own, and zk_cache_clear starts over."""
    import numpy as np
    from zkhip import _lib
    lib = _lib.load()
    _lib.check(lib.zk_cache_clear())
    s0 = _lib.cache_stats()
    omega = get_root_of_unity(8)
    vals = [FR(3 * i + 1) for i in range(8)]
    want = fft(vals, omega)
    s1 = _lib.cache_stats()
    assert s1["ntt_builds"] == s0["ntt_builds"] + 1
    for _ in range(3):
        assert fft(vals, omega) == want and ifft(want, omega) == vals
    s2 = _lib.cache_stats()
    assert s2["ntt_builds"] == s1["ntt_builds"] and s2["ntt_hits"] == s1["ntt_hits"] + 6
    fft([FR(i) for i in range(16)], get_root_of_unity(16))                  # another size: one more plan
    assert _lib.cache_stats()["ntt_builds"] == s2["ntt_builds"] + 1
    p = Polynomial([FR(1), FR(2), FR(3)])
    c0 = commit(p, srs_small)
    m1 = _lib.cache_stats()
    c1 = commit(p, srs_small)
    m2 = _lib.cache_stats()
    assert c0 == c1 and m2["msm_builds"] == m1["msm_builds"] and m2["msm_hits"] == m1["msm_hits"] + 1
    # a size class of its own (> 4096 points), twice: one build
    n = 5000
    S = _lib.ints_to_limbs([(i * 7919 + 1) % CURVE_ORDER for i in range(n)])
    K = _lib.ints_to_limbs([i + 1 for i in range(n)])
    g1 = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
    P = np.zeros((n, 8), dtype=np.uint64)
    _lib.check(lib.zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(K), n, _lib.ptr(P)))
    out, inf = np.zeros(8, dtype=np.uint64), __import__("ctypes").c_int(0)
    b0 = _lib.cache_stats()["msm_builds"]
    for _ in range(2):
        _lib.check(lib.zk_msm_g1(_lib.ptr(S), _lib.ptr(P), n, _lib.ptr(out), __import__("ctypes").byref(inf)))
    assert _lib.cache_stats()["msm_builds"] == b0 + 1
    assert _lib.limbs_to_ints(out.reshape(2, 4)) == [int(v) for v in ec_mul(G1, sum((i * 7919 + 1) * (i + 1) for i in range(n)) % CURVE_ORDER)]
    _lib.check(lib.zk_cache_clear())
    assert fft(vals, omega) == want                                          # rebuilt after the clear
    assert _lib.cache_stats()["ntt_builds"] == s2["ntt_builds"] + 2
