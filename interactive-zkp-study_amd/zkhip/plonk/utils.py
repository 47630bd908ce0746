"""Coset NTT wrappers (mirrors zkp/plonk/utils.py:145-205): evaluate on / interpolate from the
coset k*H.  The shift is folded into the GPU transform (zk_ntt_fr's coset_shift argument)."""
from ..field import FR
from .polynomial import _gpu_ntt, _log2_exact, _root_exponent


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
