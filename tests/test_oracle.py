"""CPU tests: the oracle (oracle/py_ref.py + oracle/bn254_oracle.c) against the committed golden
fixtures, the reference's surviving known-answers, and the relational identities the reference's
own tests assert.  These pin the checker before it is trusted by the GPU parity tests."""
import json
import os
import random

import numpy as np
import pytest

import c_oracle as co
import py_ref as o

P, R = o.P, o.R


def _g1(v):
    return None if v is None else (int(v[0]), int(v[1]))


def _g2(v):
    return None if v is None else ((int(v[0][0]), int(v[0][1])), (int(v[1][0]), int(v[1][1])))


@pytest.fixture(scope="module")
def toy(golden_dir):
    return json.load(open(os.path.join(golden_dir, "toy_groth16.json")))


@pytest.fixture(scope="module")
def kzg(golden_dir):
    return json.load(open(os.path.join(golden_dir, "kzg_seed42.json")))


@pytest.fixture(scope="module")
def ntt(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ntt_small.json")))


# ---------------------------------------------------------------- constants / known answers
def test_constants_appendix_a():
    assert o.g1_is_on_curve(o.G1) and o.g2_is_on_curve(o.G2)
    assert o.g1_multiply(o.G1, R) is None and o.g2_multiply(o.G2, R) is None
    assert (R - 1) % (1 << 28) == 0 and ((R - 1) >> 28) % 2 == 1  # 2-adicity 28 (zkp/plonk/field.py:171)
    assert pow(5, (R - 1) // 2, R) == R - 1                       # 5 is a non-residue
    assert o.get_root_of_unity(4) == 21888242871839275217838484774961031246007050428528088939761107053157389710902
    assert o.srs_tau(42) == 8365577799539384663899794442022354891237484320765090705979616311134436098119


def test_public_curve_constants():
    """Known answers that do not come from this repo: public constants of the curve and field the reference uses through
    py_ecc's bn128 (= Ethereum's alt_bn128).  The reference's own tests hold no EC coordinate, so
    these are the external anchors of the G1 arithmetic and of the NTT's root of unity:
      * 2*G1, the doubling of the generator (1, 2), as published with the EIP-196 ecAdd / ecMul test vectors and returned by
        py_ecc.bn128.double(G1);
      * omega_{2^28} = 5^((r-1)/2^28), the 2^28-th root of unity every BN254 toolchain ships for generator 5 (snarkjs /
        ffjavascript, gnark) -- get_root_of_unity(n) (zkp/plonk/field.py:178-180) is its 2^(28-log n)-th power."""
    two_g = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
             9918110051302171585080402603319702774565515993150576347155970296011118125764)
    assert o.g1_double(o.G1) == two_g and o.g1_add(o.G1, o.G1) == two_g and o.g1_multiply(o.G1, 2) == two_g
    assert co.g1_mul(o.G1, 2) == two_g                                   # the C oracle agrees
    w28 = 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert o.get_root_of_unity(1 << 28) == w28
    for log_n in (2, 12, 20, 22):
        assert o.get_root_of_unity(1 << log_n) == pow(w28, 1 << (28 - log_n), R)


def _sympy_curve():
    sympy = pytest.importorskip("sympy")
    from sympy.ntheory.elliptic_curve import EllipticCurve
    curve = EllipticCurve(0, 3, modulus=o.P)
    return curve, curve(1, 2)


def _sym_mul(gen, k):
    q = (k % R) * gen
    return (int(q.x), int(q.y))


def test_g1_arithmetic_against_sympy():
    """An implementation that is neither this repo's nor py_ecc's: SymPy's EllipticCurve over GF(p) (affine chord-and-tangent
    with its own modular inverses).  Multiples of the generator, sums and a doubling from both oracles must equal SymPy's --
    an independent pin of the G1 side (the reference's tests hold no EC coordinate; G2 and the pairing have no such
    second source here)."""
    curve, gen = _sympy_curve()
    rng = random.Random(17)
    ks = [1, 2, 3, 5, 35, R - 1, R - 2, (1 << 200) + 12345] + [rng.randrange(R) for _ in range(6)]
    for k in ks:
        want = _sym_mul(gen, k)
        assert o.g1_multiply(o.G1, k) == want and co.g1_mul(o.G1, k) == want, k
    a, b = rng.randrange(R), rng.randrange(R)
    pa, pb = o.g1_multiply(o.G1, a), o.g1_multiply(o.G1, b)
    s = curve(pa[0], pa[1]) + curve(pb[0], pb[1])
    assert o.g1_add(pa, pb) == (int(s.x), int(s.y)) == co.g1_add(pa, pb)
    d = curve(pa[0], pa[1]) + curve(pa[0], pa[1])
    assert o.g1_double(pa) == (int(d.x), int(d.y))
    assert o.g1_add(pa, o.g1_neg(pa)) is None


def test_golden_g1_points_against_sympy(toy, kzg, golden_dir):
    """Every G1 point of the committed fixtures that has a known discrete logarithm, recomputed with SymPy from that scalar:
    the toy Groth16 proof (A, C) and CRS, the SRS of seed 42 and its KZG commitments, and the PLONK proofs' commitments
    (= p(tau) * G1 for the polynomials stored next to them)."""
    curve, gen = _sympy_curve()
    pt = lambda v: (int(v[0]), int(v[1]))
    t = toy["inputs"]
    assert pt(toy["proof_A"]) == _sym_mul(gen, int(toy["A"])) and pt(toy["proof_C"]) == _sym_mul(gen, int(toy["C"]))
    assert [pt(p) for p in toy["sigma1_1"]] == [_sym_mul(gen, t[k]) for k in ("alpha", "beta", "delta")]
    assert [pt(p) for p in toy["sigma1_2"]] == [_sym_mul(gen, pow(t["x_val"], j, R)) for j in range(len(toy["sigma1_2"]))]
    tau = int(kzg["tau"])
    assert [pt(p) for p in kzg["g1_powers"]] == [_sym_mul(gen, pow(tau, j, R)) for j in range(len(kzg["g1_powers"]))]
    for name, case in kzg["commits"].items():
        scalar = o.horner([int(c) for c in case["coeffs"]], tau)
        assert (None if case["commitment"] is None else pt(case["commitment"])) == (None if scalar == 0 else _sym_mul(gen, scalar)), name
    with open(os.path.join(golden_dir, "plonk_proofs.json")) as f:
        plonk = json.load(f)["cases"]
    for name, case in plonk.items():
        tau = o.srs_tau(case["srs"]["seed"])
        for poly, comm in (("a_poly", "a_comm"), ("b_poly", "b_comm"), ("c_poly", "c_comm"), ("z_poly", "z_comm"), ("t_lo_poly", "t_lo_comm"),
                           ("t_mid_poly", "t_mid_comm"), ("t_hi_poly", "t_hi_comm")):
            scalar = o.horner([int(c) for c in case["polys"][poly]], tau)
            assert pt(case["proof"][comm]) == _sym_mul(gen, scalar), (name, comm)


class _SympyG2:
    """The twist y^2 = x^3 + 3/(9+i) over F_p[i]/(i^2+1) with SymPy's FiniteExtension doing ALL field arithmetic (products,
    the reduction by i^2 = -1, inversions); only the textbook chord-and-tangent formulas are written here.  A second source for
    the oracle's F_p^2 arithmetic and G2 coordinates -- not as independent as SymPy's own EllipticCurve is for G1 (the three
    formulas are restated), but none of the oracle's field code is involved."""

    def __init__(self):
        sympy = pytest.importorskip("sympy")
        from sympy.polys.agca.extensions import FiniteExtension
        self.i = sympy.symbols("i")
        self.K = FiniteExtension(sympy.Poly(self.i ** 2 + 1, self.i, domain=sympy.GF(o.P, symmetric=False)))

    def el(self, c):
        return self.K.convert(int(c[0])) + self.K.convert(int(c[1])) * self.K.convert(self.i)

    def pt(self, p):
        return None if p is None else (self.el(p[0]), self.el(p[1]))

    def coords(self, p):
        if p is None:
            return None
        out = []
        for e in p:
            lst = [int(v) % o.P for v in e.rep.to_list()]
            lst = [0] * (2 - len(lst)) + lst
            out.append((lst[1], lst[0]))
        return tuple(out)

    def dbl(self, p):
        x, y = p
        lam = (self.K.convert(3) * x * x) / (self.K.convert(2) * y)
        x3 = lam * lam - x - x
        return (x3, lam * (x - x3) - y)

    def add(self, p, q):
        if p is None or q is None:
            return q if p is None else p
        if p[0] == q[0]:
            return self.dbl(p) if p[1] == q[1] else None
        lam = (q[1] - p[1]) / (q[0] - p[0])
        x3 = lam * lam - p[0] - q[0]
        return (x3, lam * (p[0] - x3) - p[1])

    def mul(self, p, k):
        r = None
        for bit in bin(k % R)[2:]:
            r = None if r is None else self.dbl(r)
            if bit == "1":
                r = self.add(r, p)
        return r

    def on_curve(self, p):
        b2 = self.K.convert(3) / self.el((9, 1))
        return p[1] * p[1] == p[0] * p[0] * p[0] + b2


def test_g2_arithmetic_and_fixtures_against_sympy_extension_field(toy, kzg):
    g2 = _SympyG2()
    gen = g2.pt(o.G2)
    assert g2.on_curve(gen)
    rng = random.Random(23)
    for k in [2, 3, 7, R - 1] + [rng.randrange(R) for _ in range(4)]:
        want = g2.coords(g2.mul(gen, k))
        assert o.g2_multiply(o.G2, k) == want and co.g2_mul(o.G2, k) == want, k
    a, b = o.g2_multiply(o.G2, 1234567), o.g2_multiply(o.G2, 7654321)
    assert o.g2_add(a, b) == g2.coords(g2.add(g2.pt(a), g2.pt(b))) and o.g2_double(a) == g2.coords(g2.dbl(g2.pt(a)))
    assert g2.mul(gen, R) is None
    # the G2 side of the committed fixtures
    t = toy["inputs"]
    assert [_g2(p) for p in toy["sigma2_1"]] == [g2.coords(g2.mul(gen, t[k])) for k in ("beta", "gamma", "delta")]
    assert [_g2(p) for p in toy["sigma2_2"]] == [g2.coords(g2.mul(gen, pow(t["x_val"], j, R))) for j in range(len(toy["sigma2_2"]))]
    assert _g2(toy["proof_B"]) == g2.coords(g2.mul(gen, int(toy["B"])))
    assert [_g2(p) for p in kzg["g2_powers"]] == [o.G2, g2.coords(g2.mul(gen, int(kzg["tau"])))]


def test_fq12_arithmetic_against_sympy_extension_field():
    """py_ecc's FQ12 = F_p[w] / (w^12 - 18 w^6 + 82) (bn128 FQ12_MODULUS_COEFFS): products and inverses of the oracle's
    coefficient-list arithmetic against SymPy's FiniteExtension over the same modulus, and the embedding of F_p^2
    (i = w^6 - 9) that the twist uses."""
    sympy = pytest.importorskip("sympy")
    from sympy.polys.agca.extensions import FiniteExtension
    w = sympy.symbols("w")
    K = FiniteExtension(sympy.Poly(w ** 12 - 18 * w ** 6 + 82, w, domain=sympy.GF(o.P, symmetric=False)))
    gen = K.convert(w)

    def el(coeffs):
        acc, pw = K.convert(0), K.convert(1)
        for c in coeffs:
            acc = acc + K.convert(int(c)) * pw
            pw = pw * gen
        return acc

    def coeffs(e):
        lst = [int(v) % o.P for v in e.rep.to_list()]
        return ([0] * (12 - len(lst)) + lst)[::-1]

    rng = random.Random(29)
    for _ in range(3):
        a = [rng.randrange(o.P) for _ in range(12)]
        b = [rng.randrange(o.P) for _ in range(12)]
        assert [int(v) for v in o.f12_mul(a, b)] == coeffs(el(a) * el(b))
        assert [int(v) for v in o.f12_inv(a)] == coeffs(K.convert(1) / el(a))
    i_embedded = [(-9) % o.P] + [0] * 5 + [1] + [0] * 5                       # i = w^6 - 9
    assert [int(v) for v in o.f12_mul(i_embedded, i_embedded)] == [o.P - 1] + [0] * 11   # i^2 = -1


def test_ntt_against_sympy(ntt):
    """sympy.discrete.transforms.ntt(seq, prime=r) picks the smallest primitive root of r -- 5, the reference's generator
    (zkp/plonk/field.py:178-180) -- and returns natural order: the same transform as fft(coeffs, get_root_of_unity(n)).
    Both oracles and the committed NTT fixture against it, forward and inverse."""
    sympy = pytest.importorskip("sympy")
    from sympy.discrete.transforms import intt as sym_intt, ntt as sym_ntt
    assert sympy.ntheory.primitive_root(R) == 5
    rng = random.Random(31)
    for n in (1, 2, 8, 64, 1024):
        seq = [rng.randrange(R) for _ in range(n)]
        want = [int(v) for v in sym_ntt(seq, prime=R)]
        w = o.get_root_of_unity(n)
        assert o.fft(seq, w) == want
        assert co.from_limbs(co.ntt_arr(co.to_limbs(seq), w)) == want
        assert o.ifft(want, w) == seq == [int(v) for v in sym_intt(want, prime=R)]
    for name, case in ntt["cases"].items():
        coeffs = [int(c) for c in case["coeffs"]]
        assert [int(v) for v in case["fft"]] == [int(v) for v in sym_ntt(coeffs, prime=R)], name
        assert [int(v) for v in case["ifft_of_coeffs"]] == [int(v) for v in sym_intt(coeffs, prime=R)], name


def test_reference_comment_kats():
    """F_r known-answers that survive in zkp/groth16/backend.py:355,363 (with pub = [0, 1])."""
    d = o.toy_groth16()
    assert d["A"] * d["B"] % R == 21888242871839275222246405745257275088548364400416033032405666501928354297837
    assert d["VAL"][0] == 17858330771234736835653075572704017103548042849750409710240473560856989375368
    assert d["rem"] == [0, 0, 0, 0]
    assert d["Hx"][:3] == [(-528) % R, 2456, (-496) % R] and d["Hx"][3:] == [0, 0, 0, 0]


def test_val5_with_pub_0_5():
    """VAL[5] of zkp/groth16/backend.py:364 is quoted for pub = [0, 5]."""
    t = o.TOY
    Ax = [[v % R for v in row] for row in t["Ap"]]
    Bx = [[v % R for v in row] for row in t["Bp"]]
    Cx = [[v % R for v in row] for row in t["Cp"]]
    x = t["x_val"]
    Axv, Bxv, Cxv = ([o.eval_poly(p, x) for p in M] for M in (Ax, Bx, Cx))
    _, VAL = o.sigma13(6, t["alpha"], t["beta"], t["gamma"], Axv, Bxv, Cxv, [0, 5])
    assert VAL[5] == 3057428741774924004453806255227791707084339019243572619533744442206670609805


# ---------------------------------------------------------------- golden fixtures
def test_toy_groth16_golden(toy):
    d = o.toy_groth16()
    assert [str(v) for v in d["Hx"]] == toy["Hx"]
    assert _g1(toy["proof_A"]) == d["proof_A"]
    assert _g2(toy["proof_B"]) == d["proof_B"]
    assert _g1(toy["proof_C"]) == d["proof_C"]
    assert [_g1(p) for p in toy["sigma1_2"]] == d["s12"]
    assert [_g1(p) for p in toy["sigma1_4"]] == d["s14"]
    assert [_g2(p) for p in toy["sigma2_2"]] == d["s22"]
    # completeness identity of zkp/groth16/test.py:303-333
    A, B, C = int(toy["A"]), int(toy["B"]), int(toy["C"])
    t = o.TOY
    pubsum = sum(t["R"][i] * int(toy["VAL"][i]) for i in t["pub"])
    assert A * B % R == (t["alpha"] * t["beta"] + t["gamma"] * pubsum + C * t["delta"]) % R
    assert d["proof_A"] == co.g1_mul(o.G1, A) and d["proof_C"] == co.g1_mul(o.G1, C)
    assert d["proof_B"] == co.g2_mul(o.G2, B)


def test_setup_relations(toy):
    """tests/groth16/test_setup.py:15-28,37-40: sigma elements are the stated multiples of G."""
    t = o.TOY
    s11 = [_g1(p) for p in toy["sigma1_1"]]
    assert s11 == [o.g1_multiply(o.G1, t["alpha"]), o.g1_multiply(o.G1, t["beta"]), o.g1_multiply(o.G1, t["delta"])]
    s12 = [_g1(p) for p in toy["sigma1_2"]]
    assert s12[0] == o.G1 and s12[1] == o.g1_multiply(o.G1, t["x_val"])
    s13 = [_g1(p) for p in toy["sigma1_3"]]
    assert s13[2] == (0, 0) and s13[0] == o.g1_multiply(o.G1, int(toy["VAL"][0]))
    s21 = [_g2(p) for p in toy["sigma2_1"]]
    assert s21[1] == o.g2_multiply(o.G2, t["gamma"])


def test_kzg_golden(kzg):
    g1p, g2p = o.srs_generate(8, 42)
    assert [_g1(p) for p in kzg["g1_powers"]] == g1p
    assert [_g2(p) for p in kzg["g2_powers"]] == g2p
    assert _g1(kzg["two_G1"]) == (1368015179489954701390400359078579693043519447331113978918064868415326638035,
                                  9918110051302171585080402603319702774565515993150576347155970296011118125764)
    for name, c in kzg["commits"].items():
        coeffs = [int(v) for v in c["coeffs"]]
        assert o.kzg_commit(coeffs, g1p) == _g1(c["commitment"]), name
        # C oracle agrees (msm over the same bases)
        got = co.g1_from_arr(co.g1_msm_arr(co.to_limbs(coeffs), co.g1_to_arr(g1p[:len(coeffs)])))[0]
        assert got == _g1(c["commitment"]), name


def test_kzg_relations(kzg):
    """tests/plonk/test_crypto.py:113-191: constant, linear, zero, degree overflow, linearity, scaling."""
    g1p = [_g1(p) for p in kzg["g1_powers"]]
    assert o.kzg_commit([7], g1p) == o.g1_multiply(o.G1, 7)
    assert o.kzg_commit([3, 5], g1p) == o.g1_add(o.g1_multiply(g1p[0], 3), o.g1_multiply(g1p[1], 5))
    assert o.kzg_commit([0], g1p) is None
    with pytest.raises(ValueError):
        o.kzg_commit([1] * 10, g1p)
    p, q = [1, 2], [3, 4]
    assert o.kzg_commit([4, 6], g1p) == o.g1_add(o.kzg_commit(p, g1p), o.kzg_commit(q, g1p))
    assert o.kzg_commit([10, 15], g1p) == o.g1_multiply(o.kzg_commit([2, 3], g1p), 5)


def test_ntt_golden(ntt):
    assert int(ntt["omega_4"]) == o.get_root_of_unity(4)
    for name, c in ntt["cases"].items():
        coeffs = [int(v) for v in c["coeffs"]]
        w = int(c["omega"])
        exp = [int(v) for v in c["fft"]]
        assert o.fft(coeffs, w) == exp, name
        assert co.from_limbs(co.ntt_arr(co.to_limbs(coeffs), w)) == exp, name
        assert co.from_limbs(co.ntt_arr(co.to_limbs(exp), w, inverse=True)) == [v % R for v in coeffs], name
        assert o.coset_fft(coeffs, w) == [int(v) for v in c["coset_fft_k5"]], name
        assert o.ifft(coeffs, w) == [int(v) for v in c["ifft_of_coeffs"]], name


def test_fft_relations():
    """tests/plonk/test_foundation.py:486-540, 724-760."""
    assert o.fft([7], 1) == [7] and o.ifft([7], 1) == [7]
    n = 8
    w = o.get_root_of_unity(n)
    coeffs = list(range(n))
    ev = o.fft(coeffs, w)
    assert ev[0] == sum(coeffs) % R
    assert all(ev[i] == o.horner(coeffs, pow(w, i, R)) for i in range(n))
    orig = [i * 3 + 1 for i in range(n)]
    assert o.ifft(o.fft(orig, w), w) == orig and o.fft(o.ifft(orig, w), w) == orig
    assert o.coset_ifft(o.coset_fft(orig, w), w) == orig
    with pytest.raises(ValueError):
        o.get_root_of_unity(3)
    with pytest.raises(ValueError):
        o.get_root_of_unity(1 << 29)


# ---------------------------------------------------------------- C oracle vs Python oracle
def test_c_oracle_field_and_group():
    rnd = random.Random(5)
    for m, which in ((P, 0), (R, 1)):
        for _ in range(50):
            a, b = rnd.randrange(m), rnd.randrange(m)
            assert co.field_op(which, 0, a, b) == (a + b) % m
            assert co.field_op(which, 1, a, b) == (a - b) % m
            assert co.field_op(which, 2, a, b) == a * b % m
        assert co.field_op(which, 3, 12345) == pow(12345, -1, m)
    for k in (0, 1, 2, R - 1, R, rnd.randrange(R)):
        assert co.g1_mul(o.G1, k) == o.g1_multiply(o.G1, k)
        assert co.g2_mul(o.G2, k) == o.g2_multiply(o.G2, k)
    A, B = o.g1_multiply(o.G1, 123), o.g1_multiply(o.G1, 456)
    for p, q in ((A, B), (A, A), (A, o.g1_neg(A)), (None, A), (A, None), (None, None)):
        assert co.g1_add(p, q) == o.g1_add(p, q)
    A2, B2 = o.g2_multiply(o.G2, 123), o.g2_multiply(o.G2, 456)
    for p, q in ((A2, B2), (A2, A2), (A2, o.g2_neg(A2)), (None, A2)):
        assert co.g2_add(p, q) == o.g2_add(p, q)


def test_c_oracle_msm_vs_python():
    rnd = random.Random(6)
    n = 24
    sc = [rnd.randrange(R) for _ in range(n)]
    sc[3], sc[4], sc[5] = 0, 1, R - 1
    pts = [o.g1_multiply(o.G1, rnd.randrange(1, R)) for _ in range(n)]
    pts[7] = None
    pts[9] = pts[8]                 # duplicate point
    pts[11] = o.g1_neg(pts[10])     # P and -P
    sc[11] = sc[10]
    assert co.g1_from_arr(co.g1_msm_arr(co.to_limbs(sc), co.g1_to_arr(pts)))[0] == o.msm_naive(sc, pts)
    pts2 = [o.g2_multiply(o.G2, rnd.randrange(1, R)) for _ in range(6)]
    assert co.g2_from_arr(co.g2_msm_arr(co.to_limbs(sc[:6]), co.g2_to_arr(pts2)))[0] == o.msm_naive(sc[:6], pts2)
    # closed form: points k_i*G  ->  (sum s_i k_i) * G
    ks = [rnd.randrange(R) for _ in range(n)]
    P_arr = co.g1_fixed_base_arr(o.G1, co.to_limbs(ks))
    dot = co.fr_dot_arr(co.to_limbs(sc), co.to_limbs(ks))
    assert dot == sum(a * b for a, b in zip(sc, ks)) % R
    assert co.g1_from_arr(co.g1_msm_arr(co.to_limbs(sc), P_arr))[0] == o.g1_multiply(o.G1, dot)


def test_c_oracle_ntt_sizes():
    rnd = random.Random(7)
    for L in range(0, 9):
        n = 1 << L
        x = [rnd.randrange(R) for _ in range(n)]
        w = o.get_root_of_unity(n)
        assert co.from_limbs(co.ntt_arr(co.to_limbs(x), w)) == o.fft(x, w)
    x = co.to_limbs([rnd.randrange(R) for _ in range(1 << 12)])
    w = o.get_root_of_unity(1 << 12)
    y = co.ntt_arr(x, w)
    assert np.array_equal(co.ntt_arr(y, w, inverse=True), x)
    for i in (0, 1, 77, 4095):
        assert co.from_limbs(y[i:i + 1])[0] == co.fr_horner_arr(x, pow(w, i, R))


def test_c_oracle_bucket_msm_matches_naive():
    """orc_g1_msm_bucket (serial Pippenger) == orc_g1_msm (per-term double-and-add) for several window widths."""
    rng = np.random.default_rng(77)
    n = 200
    sc = [int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, o.R - 1, 1
    ks = [int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(n)]
    P = co.g1_fixed_base_arr(o.G1, co.to_limbs(ks))
    P[5] = 0                      # infinity input
    P[7] = P[6]                   # duplicate point
    S = co.to_limbs(sc)
    want = co.g1_msm_arr(S, P)
    for c in (1, 4, 11, 16):
        assert (co.g1_msm_bucket_arr(S, P, c) == want).all()
    assert (co.g1_msm_bucket_arr(S[:0], P[:0], 8) == 0).all()
    for threads in (1, 3, 64):
        assert (co.g1_msm_bucket_mt_arr(S, P, 7, threads) == want).all()   # windows spread over threads


def test_c_oracle_g2_bucket_msm_matches_naive():
    """orc_g2_msm_bucket (serial Pippenger in G2) == orc_g2_msm (per-term double-and-add), several window widths; the G2
    fixed-base batch is the per-scalar orc_g2_mul."""
    rng = np.random.default_rng(78)
    n = 120
    sc = [int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, o.R - 1, 1
    ks = [int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(n)]
    P = co.g2_fixed_base_arr(o.G2, co.to_limbs(ks))
    assert co.g2_from_arr(P[3])[0] == co.g2_mul(o.G2, ks[3])
    P[5] = 0                      # infinity input
    P[7] = P[6]                   # duplicate point
    S = co.to_limbs(sc)
    want = co.g2_msm_arr(S, P)
    for c in (1, 5, 13):
        assert (co.g2_msm_bucket_arr(S, P, c) == want).all()
    assert (co.g2_msm_bucket_arr(S[:0], P[:0], 8) == 0).all()
    # closed form: P_i = k_i G2  =>  MSM = (sum s_i k_i) G2
    sc2 = list(sc)
    ks2 = list(ks)
    ks2[5] = 0
    ks2[7] = ks2[6]
    assert co.g2_from_arr(want)[0] == co.g2_mul(o.G2, sum(a * b for a, b in zip(sc2, ks2)) % o.R)
