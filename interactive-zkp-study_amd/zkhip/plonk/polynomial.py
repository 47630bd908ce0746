"""Polynomials over F_r and the NTT entry points (mirrors zkp/plonk/polynomial.py).

fft / ifft keep the reference's signatures and conventions (natural order in and out, omega
supplied by the caller, ifft = fft with omega^-1 then * n^-1; polynomial.py:292-378) and run
in libzkhip's multi-pass LDS NTT.  The GPU transform uses omega_n = 5^((r-1)/n), which is what
every reference call site passes (get_root_of_unity, zkp/plonk/field.py:178-180); any other
primitive n-th root w = omega_n^e is served by the same kernel through the index map
X_w[k] = X_omega[(e*k) mod n].
"""
import numpy as np

from .. import _lib
from ..field import FR, CURVE_ORDER, get_root_of_unity


class Polynomial:
    """Coefficient-form polynomial over F_r with the reference's normalisation
    (trailing zeros trimmed, zero polynomial = [0] with degree 0; polynomial.py:45-83)."""

    def __init__(self, coeffs=None):
        if coeffs is None:
            self.coeffs = [FR(0)]
        else:
            self.coeffs = [c if isinstance(c, FR) else FR(c) for c in coeffs]
            if not self.coeffs:
                self.coeffs = [FR(0)]
        self._trim()

    def _trim(self):
        while len(self.coeffs) > 1 and self.coeffs[-1] == FR(0):
            self.coeffs.pop()

    @property
    def degree(self):
        if len(self.coeffs) == 1 and self.coeffs[0] == FR(0):
            return 0
        return len(self.coeffs) - 1

    def is_zero(self):
        return len(self.coeffs) == 1 and self.coeffs[0] == FR(0)

    def evaluate(self, point):
        """Horner evaluation (polynomial.py:85-106)."""
        x = int(point) % CURVE_ORDER
        acc = 0
        for c in reversed(self.coeffs):
            acc = (acc * x + c.n) % CURVE_ORDER
        return FR(acc)

    def __add__(self, other):
        if isinstance(other, (int, FR)):
            other = Polynomial([other])
        n = max(len(self.coeffs), len(other.coeffs))
        a = self.coeffs + [FR(0)] * (n - len(self.coeffs))
        b = other.coeffs + [FR(0)] * (n - len(other.coeffs))
        return Polynomial([x + y for x, y in zip(a, b)])

    __radd__ = __add__

    def __neg__(self):
        return Polynomial([-c for c in self.coeffs])

    def __sub__(self, other):
        if isinstance(other, (int, FR)):
            other = Polynomial([other])
        return self + (-other)

    def __rsub__(self, other):
        if isinstance(other, (int, FR)):
            other = Polynomial([other])
        return other.__sub__(self)

    def __mul__(self, other):
        if isinstance(other, (int, FR)):
            s = FR(other)
            return Polynomial([c * s for c in self.coeffs])
        out = [0] * (len(self.coeffs) + len(other.coeffs) - 1)
        for i, a in enumerate(self.coeffs):
            if a.n == 0:
                continue
            for j, b in enumerate(other.coeffs):
                out[i + j] += a.n * b.n
        return Polynomial([v % CURVE_ORDER for v in out])

    __rmul__ = __mul__

    def __eq__(self, other):
        if isinstance(other, (int, FR)):
            other = Polynomial([other])
        if not isinstance(other, Polynomial):
            return False
        return self.coeffs == other.coeffs

    __hash__ = None

    def __len__(self):
        return len(self.coeffs)

    def scale(self, scalar):
        """scalar * p(x) (polynomial.py:189-198)."""
        return self * scalar

    def divide_by_vanishing(self, n):
        """p(x) / (x^n - 1); ValueError unless the division is exact (polynomial.py:200-224)."""
        q, r = poly_div(self, Polynomial.vanishing(n))
        if not r.is_zero():
            raise ValueError("not divisible by the vanishing polynomial (constraints unsatisfied)")
        return q

    @staticmethod
    def _vanishing_coeffs(n):
        return Polynomial([FR(CURVE_ORDER - 1)] + [FR(0)] * (n - 1) + [FR(1)])

    @classmethod
    def one(cls):
        return cls([FR(1)])

    @classmethod
    def vanishing(cls, n):
        """Z_H(x) = x^n - 1 (polynomial.py:243-261)."""
        return cls._vanishing_coeffs(n)

    def __repr__(self):
        return "Polynomial(%r)" % ([c.n for c in self.coeffs],)

    @classmethod
    def zero(cls):
        return cls([FR(0)])

    @classmethod
    def from_evaluations(cls, evals, omega):
        """Interpolate over {omega^i} by inverse NTT (polynomial.py:263-285)."""
        return cls(ifft(evals, omega))


def _log2_exact(n):
    log_n = n.bit_length() - 1
    if n < 1 or (1 << log_n) != n:
        raise ValueError("fft length must be a power of two, got %d" % n)
    return log_n


def _root_exponent(omega, n):
    """e with omega == omega_n^e, or raises when omega is not a primitive n-th root of unity."""
    w = int(omega) % CURVE_ORDER
    base = int(get_root_of_unity(n))
    if w == base:
        return 1
    # n <= 2^28 and the group is cyclic of 2-power order: solve bit by bit (Pohlig-Hellman).
    log_n = n.bit_length() - 1
    if pow(w, n, CURVE_ORDER) != 1 or (n > 1 and pow(w, n // 2, CURVE_ORDER) == 1):
        raise ValueError("omega is not a primitive %d-th root of unity" % n)
    e, base_inv = 0, pow(base, -1, CURVE_ORDER)
    for b in range(log_n):
        t = w * pow(base_inv, e, CURVE_ORDER) % CURVE_ORDER
        if pow(t, n >> (b + 1), CURVE_ORDER) != 1:
            e |= 1 << b
    assert pow(base, e, CURVE_ORDER) == w
    return e


def _gpu_ntt(values, inverse, coset_shift=None):
    n = len(values)
    log_n = _log2_exact(n)
    arr = _lib.ints_to_limbs([int(v) % CURVE_ORDER for v in values])
    k = None if coset_shift is None else _lib.ints_to_limbs([int(coset_shift) % CURVE_ORDER])
    _lib.check(_lib.load().zk_ntt_fr(_lib.ptr(arr), log_n, 1 if inverse else 0, None if k is None else _lib.ptr(k)))
    return _lib.limbs_to_ints(arr)


def fft(coeffs, omega):
    """Evaluate at {1, omega, ..., omega^(n-1)} (polynomial.py:292-341)."""
    n = len(coeffs)
    _log2_exact(n)
    e = _root_exponent(omega, n)
    out = _gpu_ntt(coeffs, inverse=False)
    if e != 1:
        out = [out[(e * k) % n] for k in range(n)]
    return [FR(v) for v in out]


def ifft(evals, omega):
    """Inverse of fft: interpolate (polynomial.py:344-378)."""
    n = len(evals)
    _log2_exact(n)
    e = _root_exponent(omega, n)
    vals = [int(v) % CURVE_ORDER for v in evals]
    if e != 1:  # evals[k] = X_omega_n[(e*k) mod n]  ->  undo the index map first
        nat = [0] * n
        for k in range(n):
            nat[(e * k) % n] = vals[k]
        vals = nat
    out = _gpu_ntt(vals, inverse=True)
    return [FR(v) for v in out]


def poly_div(a, b):
    """Long division a = b*q + r (polynomial.py:385-435); scalar host glue."""
    if b.is_zero():
        raise ValueError("division by the zero polynomial")
    rem = [c.n for c in a.coeffs]
    div = [c.n for c in b.coeffs]
    deg_b, deg_a = len(div) - 1, len(rem) - 1
    if deg_a < deg_b:
        return Polynomial.zero(), Polynomial(rem)
    quot = [0] * (deg_a - deg_b + 1)
    lead_inv = pow(div[-1], -1, CURVE_ORDER)
    for i in range(deg_a - deg_b, -1, -1):
        coeff = rem[i + deg_b] * lead_inv % CURVE_ORDER
        quot[i] = coeff
        if coeff:
            for j in range(deg_b + 1):
                rem[i + j] = (rem[i + j] - coeff * div[j]) % CURVE_ORDER
    return Polynomial(quot), Polynomial(rem)


def lagrange_basis(domain, i):
    """L_i(x) = prod_{j != i} (x - d_j) / (d_i - d_j) in coefficient form (polynomial.py:438-475); scalar host glue
    (the reference's round 3 uses it for L_1; the backend's round 3 evaluates L_1 in closed form on the coset)."""
    result = Polynomial([FR(1)])
    denominator = FR(1)
    for j, dj in enumerate(domain):
        if j == i:
            continue
        result = result * Polynomial([FR(0) - dj, FR(1)])
        denominator = denominator * (domain[i] - dj)
    return result * (FR(1) / denominator)
