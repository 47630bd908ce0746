"""Pieces shared by the PLONK round modules and the verifier."""
from ...field import FR
from ..permutation import K1, K2
from ..polynomial import Polynomial

COSET_K = 5
# round3.py:140-147 raises ValueError with this (Korean) text when C(x) is not a multiple of Z_H(x); the reference's own test
# matches on it (tests/plonk/test_prover.py:746), so the text is part of the error behaviour.  English gloss appended.
NOT_DIVISIBLE = ("제약 다항식이 Z_H(x)로 나누어 떨어지지 않습니다. 회로 또는 witness에 오류가 있습니다. "
                 "(the constraint polynomial is not divisible by Z_H(x): circuit or witness is inconsistent)")


def pad_rows(vals, n):
    vals = list(vals)
    return vals + [FR(0)] * (n - len(vals))


def times_vanishing(blind, n):
    """blind(x) * (x^n - 1)."""
    coeffs = [FR(0)] * (n + len(blind))
    for i, b in enumerate(blind):
        coeffs[i] = coeffs[i] - b
        coeffs[n + i] = coeffs[n + i] + b
    return Polynomial(coeffs)


def linearisation_scalars(alpha, beta, gamma, zeta, n, omega, a_eval, b_eval, c_eval, s1_eval, s2_eval, z_omega_eval):
    """Scalars shared by round 5 and the verifier (round5.py:92-132, verifier.py:103-131)."""
    zh_zeta = zeta ** n - FR(1)
    l1_zeta = FR(1) if zeta == FR(1) else zh_zeta / (FR(n) * (zeta - FR(1)))
    perm_z = alpha * (a_eval + beta * zeta + gamma) * (b_eval + beta * K1 * zeta + gamma) * (c_eval + beta * K2 * zeta + gamma)
    ab = (a_eval + beta * s1_eval + gamma) * (b_eval + beta * s2_eval + gamma)
    perm_s3 = alpha * ab * beta * z_omega_eval
    r0 = FR(0) - alpha * ab * z_omega_eval * (c_eval + gamma) - alpha * alpha * l1_zeta
    return zh_zeta, l1_zeta, perm_z, perm_s3, r0
