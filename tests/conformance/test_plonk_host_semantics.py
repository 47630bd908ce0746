"""Conformance of the host-side value types and bookkeeping with what the reference's own test-suite expects
(tests/plonk/test_foundation.py: FR, roots of unity, Polynomial, poly_div, lagrange_basis, the closed-form helpers, padding;
tests/plonk/test_circuit.py: Gate, Circuit, copy constraints, permutation labels, accumulator, Transcript).

These are the behaviours a caller of zkp.plonk observes without touching the curve or the NTT, so they run in the CPU suite.
The cases are the reference's (same inputs, same expected values), restated as tables; everything that needs the backend
(EC operations, fft, commit, the prover) is in test_plonk_backend.py of this directory."""
import hashlib

import pytest

from zkhip.field import CURVE_ORDER, FR, Z1, get_root_of_unity, get_roots_of_unity
from zkhip.plonk.circuit import Circuit, Gate
from zkhip.plonk.permutation import K1, K2, build_permutation_polynomials, compute_accumulator
from zkhip.plonk.polynomial import Polynomial, lagrange_basis, poly_div
from zkhip.plonk.transcript import Transcript
from zkhip.plonk.utils import lagrange_basis_eval, next_power_of_2, pad_to_power_of_2, public_input_poly_eval, vanishing_poly_eval

R = CURVE_ORDER
P = Polynomial


# ------------------------------------------------------------------ FR (test_foundation.py:27-100)
@pytest.mark.parametrize("expr,want", [
    (lambda: int(FR(0)), 0), (lambda: int(FR(1)), 1), (lambda: int(FR(R - 1)), R - 1),
    (lambda: FR(R), FR(0)), (lambda: FR(R + 7), FR(7)),
    (lambda: FR(3) + FR(5), FR(8)), (lambda: FR(R - 1) + FR(2), FR(1)),
    (lambda: FR(10) - FR(3), FR(7)), (lambda: FR(0) - FR(1), FR(R - 1)),
    (lambda: FR(6) * FR(7), FR(42)), (lambda: FR(12345) * FR(0), FR(0)),
    (lambda: FR(42) / FR(7), FR(6)), (lambda: FR(3) * (FR(1) / FR(3)), FR(1)),
    (lambda: FR(2) ** 10, FR(1024)), (lambda: FR(7) ** (R - 1), FR(1)),
    (lambda: FR(5) + (FR(0) - FR(5)), FR(0)),
])
def test_fr_arithmetic(expr, want):
    assert expr() == want


def test_fr_equality_modulus_and_identity_point():
    assert FR(3) == FR(3) and FR(3) != FR(4)
    assert FR.field_modulus == R
    assert Z1 is None                                   # test_foundation.py:172


# ------------------------------------------------------------------ roots of unity (test_foundation.py:180-226)
def test_roots_of_unity():
    assert get_root_of_unity(1) == FR(1)
    for k in (2, 4, 8, 16):
        assert get_root_of_unity(k) ** k == FR(1)
    w4 = get_root_of_unity(4)
    assert w4 ** 4 == FR(1) and w4 ** 2 != FR(1) and w4 != FR(1)
    for bad in (3, 0, 1 << 29):
        with pytest.raises(ValueError):
            get_root_of_unity(bad)
    roots = get_roots_of_unity(8)
    assert len(roots) == 8 and get_roots_of_unity(4)[0] == FR(1)
    assert all(r ** 8 == FR(1) for r in roots) and len({int(r) for r in roots}) == 8


# ------------------------------------------------------------------ Polynomial (test_foundation.py:231-458)
def test_polynomial_construction_and_shape():
    assert P([FR(1), FR(2), FR(3)]).coeffs == [FR(1), FR(2), FR(3)] == P([1, 2, 3]).coeffs
    assert P().is_zero()
    assert len(P([FR(1), FR(2), FR(0), FR(0)]).coeffs) == 2
    assert [P([FR(5)]).degree, P([FR(1), FR(2)]).degree, P([FR(1), FR(0), FR(3)]).degree, P.zero().degree] == [0, 1, 2, 0]
    assert P.zero().is_zero() and P([FR(0)]).is_zero() and not P([FR(1)]).is_zero() and not P([FR(0), FR(1)]).is_zero()
    assert len(P([FR(1), FR(2), FR(3)])) == 3
    assert "Poly" in repr(P([FR(1), FR(2)]))
    assert P.zero().coeffs == [FR(0)]
    assert P.one().coeffs == [FR(1)] and not P.one().is_zero()


@pytest.mark.parametrize("expr,want", [
    (lambda: P([1, 2]) + P([3, 4]), [4, 6]),
    (lambda: P([1, 2, 3]) + P([4]), [5, 2, 3]),
    (lambda: P([1, 2]) + FR(3), [4, 2]), (lambda: P([1, 2]) + 3, [4, 2]), (lambda: 3 + P([1, 2]), [4, 2]),
    (lambda: P([5, 7]) - P([2, 3]), [3, 4]),
    (lambda: 5 - P([1, 2]), [4, R - 2]),
    (lambda: -P([1, 2]), [R - 1, R - 2]),
    (lambda: P([1, 2]) * P([3, 4]), [3, 10, 8]),
    (lambda: P([1, 2]) * FR(3), [3, 6]), (lambda: P([1, 2]) * 3, [3, 6]), (lambda: 3 * P([1, 2]), [3, 6]),
    (lambda: P([1, 2]).scale(FR(3)), [3, 6]),
])
def test_polynomial_arithmetic(expr, want):
    assert expr().coeffs == [FR(v) for v in want]


def test_polynomial_comparisons_and_evaluation():
    assert P([1, 2]) == P([1, 2]) and P([1, 2]) != P([1, 3]) and P([FR(5)]) == 5
    assert (P([1, 2]) + (-P([1, 2]))).is_zero()
    for coeffs, x, want in (([7], FR(100), 7), ([3, 2], FR(5), 13), ([1, 2, 3], FR(2), 17), ([0], FR(42), 0), ([5, 3, 2], FR(0), 5), ([1, 1], 4, 5)):
        assert P(coeffs).evaluate(x) == FR(want)


def test_vanishing_polynomial_and_exact_division():
    zh = P.vanishing(4)
    assert all(zh.evaluate(r) == FR(0) for r in get_roots_of_unity(4))
    assert zh.degree == 4 and zh.coeffs[-1] == FR(1) and zh.coeffs[0] == FR(R - 1)
    factor = P([FR(1), FR(1)])
    assert (zh * factor).divide_by_vanishing(4) == factor
    with pytest.raises(ValueError):
        P([FR(1), FR(0), FR(1)]).divide_by_vanishing(4)                     # x^2 + 1 is not a multiple of x^4 - 1


def test_poly_div():                                                          # test_foundation.py:547-586
    q, r = poly_div(P([R - 1, 0, 1]), P([R - 1, 1]))                         # (x^2 - 1) / (x - 1)
    assert q == P([1, 1]) and r.is_zero()
    a, b = P([1, 0, 1]), P([R - 1, 1])
    q, r = poly_div(a, b)
    assert b * q + r == a
    q, r = poly_div(P([3]), P([1, 1]))
    assert q.is_zero() and r == P([3])
    with pytest.raises(ValueError):
        poly_div(P([1, 2]), P.zero())
    a, b = P([5, 3, 7, 2]), P([1, 1])
    q, r = poly_div(a, b)
    assert b * q + r == a


def test_lagrange_basis_and_closed_forms():                                    # test_foundation.py:590-671
    for domain in ([FR(1), FR(2), FR(3)], get_roots_of_unity(4)):
        for i in range(len(domain)):
            li = lagrange_basis(domain, i)
            assert [li.evaluate(d) for d in domain] == [FR(1) if j == i else FR(0) for j in range(len(domain))]
    n, w, roots = 4, get_root_of_unity(4), get_roots_of_unity(4)
    assert all(vanishing_poly_eval(n, w ** i) == FR(0) for i in range(n))
    assert vanishing_poly_eval(4, FR(17)) == FR(17) ** 4 - FR(1)
    assert vanishing_poly_eval(n, FR(42)) == P.vanishing(n).evaluate(FR(42))
    for i in range(n):
        assert [lagrange_basis_eval(i, n, w, w ** j) for j in range(n)] == [FR(1) if j == i else FR(0) for j in range(n)]
        assert lagrange_basis_eval(i, n, w, FR(17)) == lagrange_basis(roots, i).evaluate(FR(17))
    total = FR(0)
    for i in range(n):
        total = total + lagrange_basis_eval(i, n, w, FR(99))
    assert total == FR(1)
    assert public_input_poly_eval([], n, w, FR(7)) == FR(0)


def test_padding_helpers():                                                   # test_foundation.py:767-810
    assert [next_power_of_2(k) for k in (1, 2, 4, 8, 3, 5, 7, 9, 0)] == [1, 2, 4, 8, 4, 8, 8, 16, 1]
    assert len(pad_to_power_of_2([FR(1), FR(2), FR(3), FR(4)])) == 4
    padded = pad_to_power_of_2([FR(1), FR(2), FR(3)])
    assert len(padded) == 4 and padded[3] == FR(0)
    assert pad_to_power_of_2([FR(1), FR(2), FR(3)], fill=FR(99))[3] == FR(99)
    assert pad_to_power_of_2([FR(42)]) == [FR(42)]
    assert pad_to_power_of_2([]) == [FR(0)]


# ------------------------------------------------------------------ Gate / Circuit (test_circuit.py:27-400)
def test_gate_equation():
    g = Gate(FR(1), FR(2), FR(3), FR(4), FR(5))
    assert (g.q_l, g.q_r, g.q_o, g.q_m, g.q_c) == (FR(1), FR(2), FR(3), FR(4), FR(5))
    g = Gate(1, 2, 3, 4, 5)
    assert (g.q_l, g.q_c) == (FR(1), FR(5))
    mul, add, const = Gate(0, 0, R - 1, 1, 0), Gate(1, 1, R - 1, 0, 0), Gate(1, 0, R - 1, 0, 5)
    assert mul.check(FR(3), FR(5), FR(15)) is True and mul.check(FR(3), FR(5), FR(16)) is False
    assert add.check(FR(3), FR(5), FR(8)) is True and add.check(FR(3), FR(5), FR(9)) is False
    assert const.check(FR(30), FR(0), FR(35)) is True and const.check(FR(30), FR(0), FR(36)) is False
    assert mul.check(3, 5, 15) is True                                       # int inputs
    assert Gate(0, 0, 0, 0, 0).check(FR(99), FR(7), FR(1)) is True            # the padding gate accepts anything
    assert Gate(2, 3, R - 1, 1, 4).check(FR(5), FR(7), FR(2 * 5 + 3 * 7 + 5 * 7 + 4)) is True


def test_circuit_builders():
    c = Circuit()
    assert c.n == 0 and c.gates == [] and c.copy_constraints == [] and c.num_public_inputs == 0
    assert all(len(col) == 0 for col in c.get_selector_polynomials())
    assert [c.add_multiplication_gate(), c.add_addition_gate(), c.add_constant_gate(5), c.add_public_input_gate()] == [0, 1, 2, 3]
    assert c.n == 4 and c.num_public_inputs == 1
    sel = lambda g: (g.q_l, g.q_r, g.q_o, g.q_m, g.q_c)
    assert sel(c.gates[0]) == (FR(0), FR(0), FR(R - 1), FR(1), FR(0))
    assert sel(c.gates[1]) == (FR(1), FR(1), FR(R - 1), FR(0), FR(0))
    assert sel(c.gates[2]) == (FR(1), FR(0), FR(R - 1), FR(0), FR(5))
    assert sel(c.gates[3]) == (FR(0), FR(0), FR(1), FR(0), FR(0))
    c.add_public_input_gate()
    assert c.num_public_inputs == 2
    d = Circuit()
    d.add_constant_gate(FR(42))
    assert d.gates[0].q_c == FR(42)
    q_l, q_r, q_o, q_m, q_c = c.get_selector_polynomials()
    assert len(q_l) == 5 and q_m[0] == FR(1) and q_o[0] == FR(R - 1) and (q_l[1], q_r[1], q_m[1]) == (FR(1), FR(1), FR(0)) and q_c[2] == FR(5)
    with pytest.raises(NotImplementedError):
        c.compute_witness({"x": FR(3)})


def test_copy_constraints_to_permutation():
    c = Circuit()
    c.add_multiplication_gate()
    c.add_addition_gate()
    assert c.build_copy_constraints() == list(range(6))                      # no constraints: identity
    c = Circuit()
    c.add_multiplication_gate()
    c.add_multiplication_gate()
    c.add_copy_constraint(0, 2, 1, 0)                                        # gate0.c (position 2*2+0) == gate1.a (position 0*2+1)
    assert c.copy_constraints == [(0, 2, 1, 0)]
    assert c.build_copy_constraints() == [0, 4, 2, 3, 1, 5]
    c = Circuit()
    for _ in range(2):
        c.add_multiplication_gate()
    c.add_addition_gate()
    c.add_copy_constraint(0, 0, 0, 1)
    sigma = c.build_copy_constraints()
    assert sigma[0] == 3 and sigma[3] == 0


def test_example_circuit():
    out = Circuit.x3_plus_x_plus_5_eq_35()
    assert len(out) == 5
    circuit, a, b, c, pub = out
    assert circuit.n == 4 and pub == [FR(35)] and circuit.num_public_inputs == 1 and len(circuit.copy_constraints) == 6
    assert (a, b, c) == ([FR(3), FR(9), FR(27), FR(30)], [FR(3), FR(3), FR(3), FR(0)], [FR(9), FR(27), FR(30), FR(35)])
    assert all(g.check(a[i], b[i], c[i]) for i, g in enumerate(circuit.gates))
    wires = a + b + c
    sigma = circuit.build_copy_constraints()
    assert sorted(sigma) == list(range(12)) and all(wires[i] == wires[sigma[i]] for i in range(12))
    assert (circuit.gates[0].q_m, circuit.gates[2].q_l, circuit.gates[3].q_c) == (FR(1), FR(1), FR(5))


# ------------------------------------------------------------------ permutation (test_circuit.py:405-620)
def test_permutation_labels_and_accumulator():
    n = 4
    dom = get_roots_of_unity(n)
    s1, s2, s3 = build_permutation_polynomials(list(range(3 * n)), n, dom)
    assert s1 == dom and s2 == [K1 * d for d in dom] and s3 == [K2 * d for d in dom]
    assert (K1, K2) == (FR(2), FR(3))
    sigma = list(range(12))
    sigma[0], sigma[4] = sigma[4], sigma[0]
    s1, s2, s3 = build_permutation_polynomials(sigma, n, dom)
    assert s1[0] == K1 * dom[0] and s2[0] == dom[0]
    circuit, a, b, c, _ = Circuit.x3_plus_x_plus_5_eq_35()
    sigma = circuit.build_copy_constraints()
    cols = build_permutation_polynomials(sigma, n, dom)
    assert all(len(col) == n for col in cols)
    coset = set(int(k * d) for k in (FR(1), K1, K2) for d in dom)
    assert all(int(v) in coset for col in cols for v in col)
    beta, gamma = FR(31), FR(47)                                             # the reference's fixture values
    z = compute_accumulator(a, b, c, sigma, n, dom, beta, gamma)
    assert len(z) == n and z[0] == FR(1)
    # the product closes: z_{n-1} * num_{n-1} / den_{n-1} == 1 (test_circuit.py:539-560)
    i = n - 1
    num = (a[i] + beta * dom[i] + gamma) * (b[i] + beta * K1 * dom[i] + gamma) * (c[i] + beta * K2 * dom[i] + gamma)
    den = (a[i] + beta * cols[0][i] + gamma) * (b[i] + beta * cols[1][i] + gamma) * (c[i] + beta * cols[2][i] + gamma)
    assert z[i] * num / den == FR(1)
    assert compute_accumulator(a, b, c, sigma, n, dom, FR(100), FR(200)) != z
    ident = compute_accumulator([FR(v) for v in (1, 2, 3, 4)], [FR(v) for v in (5, 6, 7, 8)], [FR(v) for v in (9, 10, 11, 12)],
                                list(range(12)), n, dom, FR(13), FR(17))
    assert ident == [FR(1)] * n                                              # identity permutation: every factor is 1
    beta, gamma = FR(7), FR(11)
    bad_a = list(a)
    bad_a[0] = FR(999)                                                       # breaks the copies of x: the product no longer closes
    zb = compute_accumulator(bad_a, b, c, sigma, n, dom, beta, gamma)
    numb = (bad_a[i] + beta * dom[i] + gamma) * (b[i] + beta * K1 * dom[i] + gamma) * (c[i] + beta * K2 * dom[i] + gamma)
    denb = (bad_a[i] + beta * cols[0][i] + gamma) * (b[i] + beta * cols[1][i] + gamma) * (c[i] + beta * cols[2][i] + gamma)
    assert zb[i] * numb / denb != FR(1)
    # py_ecc convention: FR(x) / FR(0) == FR(0).  With beta = 7, gamma = 11 the honest witness hits a zero denominator in row 0
    # (3 + 7 * (2 * omega^2) + 11 = 0), and the reference's loop then carries z = 0 from row 1 on
    assert FR(5) / FR(0) == FR(0) and FR(0) / FR(0) == FR(0)
    zz = compute_accumulator(a, b, c, sigma, n, dom, beta, gamma)
    assert zz == [FR(1), FR(0), FR(0), FR(0)]


# ------------------------------------------------------------------ transcript (test_circuit.py:620-786)
def test_transcript():
    from zkhip.field import FQ
    g1 = (FQ(1), FQ(2))
    t1, t2 = Transcript(), Transcript()
    for t in (t1, t2):
        t.append_scalar(b"x", FR(42))
    c1 = t1.challenge_scalar(b"c")
    assert c1 == t2.challenge_scalar(b"c") and isinstance(c1, FR) and c1 != FR(0) and 0 <= int(c1) < R
    a, b = Transcript(), Transcript()
    a.append_scalar(b"x", FR(1))
    b.append_scalar(b"x", FR(2))
    assert a.challenge_scalar(b"c") != b.challenge_scalar(b"c")
    a, b = Transcript(), Transcript()
    assert a.challenge_scalar(b"alpha") != b.challenge_scalar(b"beta")        # labels separate domains
    t = Transcript()
    before = len(t.state)
    first = t.challenge_scalar(b"c")
    assert len(t.state) == before + 1 + 32 and first != t.challenge_scalar(b"c")   # the digest is chained into the state
    t = Transcript()
    t.append_scalar(b"s", FR(7))
    assert bytes(t.state) == b"plonk" + b"s" + (7).to_bytes(32, "big")
    t = Transcript()
    t.append_point(b"p", g1)
    assert bytes(t.state) == b"plonk" + b"p" + (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
    n1, n2 = Transcript(), Transcript()
    n1.append_point(b"p", None)
    n2.append_point(b"p", None)
    assert bytes(n1.state) == b"plonk" + b"p" + bytes(64) and n1.challenge_scalar(b"c") == n2.challenge_scalar(b"c")
    t = Transcript()
    t.append_point(b"p", g1)
    assert t.challenge_scalar(b"c") != n1.challenge_scalar(b"c2")
    assert bytes(Transcript(label=b"custom").state) == b"custom" and bytes(Transcript().state) == b"plonk"
    x, y = Transcript(), Transcript()
    x.append_scalar(b"a", FR(1)); x.append_scalar(b"b", FR(2))
    y.append_scalar(b"b", FR(2)); y.append_scalar(b"a", FR(1))
    assert x.challenge_scalar(b"c") != y.challenge_scalar(b"c")              # order matters
    t = Transcript()
    t.append_scalar(b"v", FR(R + 5))                                         # stored reduced
    assert bytes(t.state)[-32:] == (5).to_bytes(32, "big")
    want = int.from_bytes(hashlib.sha256(b"plonk" + b"beta").digest(), "big") % R
    assert int(Transcript().challenge_scalar(b"beta")) == want
