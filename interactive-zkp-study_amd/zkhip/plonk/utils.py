"""PLONK helpers (mirrors zkp/plonk/utils.py): the closed-form evaluations the prover and verifier share
(vanishing_poly_eval, lagrange_basis_eval, public-input polynomial: utils.py:25-141), the coset NTT wrappers
(utils.py:145-205: evaluate on / interpolate from the coset k*H; the shift is folded into the GPU transform, zk_ntt_fr's
coset_shift argument) and the padding helpers (utils.py:208-246)."""
from ..field import FR
from .polynomial import Polynomial, _gpu_ntt, _log2_exact, _root_exponent


def vanishing_poly_eval(n, zeta):
    """Z_H(zeta) = zeta^n - 1 (utils.py:25-43)."""
    if not isinstance(zeta, FR):
        zeta = FR(zeta)
    return zeta ** n - FR(1)


def lagrange_basis_eval(i, n, omega, zeta):
    """L_i(zeta) = (omega^i / n) (zeta^n - 1) / (zeta - omega^i); 1 at zeta = omega^i (utils.py:46-80)."""
    if not isinstance(zeta, FR):
        zeta = FR(zeta)
    omega_i = FR(omega) ** i
    denominator = zeta - omega_i
    if denominator == FR(0):
        return FR(1)
    return FR(1) / FR(n) * vanishing_poly_eval(n, zeta) * omega_i / denominator


def public_input_polynomial(pub_inputs, n, omega):
    """PI(x) = sum_i w_i L_i(x): the inverse NTT of [w_0, w_1, ..., 0, ...] (utils.py:83-116)."""
    if not pub_inputs:
        return Polynomial.zero()
    evals = [FR(0)] * n
    for i, val in enumerate(pub_inputs):
        evals[i] = val if isinstance(val, FR) else FR(val)
    return Polynomial.from_evaluations(evals, omega)


def public_input_poly_eval(pub_inputs, n, omega, zeta):
    """PI(zeta) from the Lagrange evaluations alone (utils.py:119-141)."""
    result = FR(0)
    for i, val in enumerate(pub_inputs):
        result = result + (val if isinstance(val, FR) else FR(val)) * lagrange_basis_eval(i, n, omega, zeta)
    return result


def coset_fft(coeffs, omega, k=None):
    """fft of [c_i * k^i]  (utils.py:145-176; default k = FR(5))."""
    if k is None:
        k = FR(5)
    n = len(coeffs)
    _log2_exact(n)
    e = _root_exponent(omega, n)
    out = _gpu_ntt(coeffs, inverse=False, coset_shift=k)
    if e != 1:
        out = [out[(e * i) % n] for i in range(n)]
    return [FR(v) for v in out]


def coset_ifft(evals, omega, k=None):
    """ifft, then c_i * k^-i  (utils.py:179-205)."""
    if k is None:
        k = FR(5)
    n = len(evals)
    _log2_exact(n)
    e = _root_exponent(omega, n)
    vals = [int(v) for v in evals]
    if e != 1:
        nat = [0] * n
        for i in range(n):
            nat[(e * i) % n] = vals[i]
        vals = nat
    return [FR(v) for v in _gpu_ntt(vals, inverse=True, coset_shift=k)]


def next_power_of_2(n):
    """Smallest power of two >= n (utils.py:227-246)."""
    if n <= 1:
        return 1
    p = 1
    while p < n:
        p <<= 1
    return p


def pad_to_power_of_2(lst, fill=None):
    """List padded with `fill` (default FR(0)) to a power-of-two length (utils.py:208-224)."""
    if fill is None:
        fill = FR(0)
    return list(lst) + [fill] * (next_power_of_2(len(lst)) - len(lst))
