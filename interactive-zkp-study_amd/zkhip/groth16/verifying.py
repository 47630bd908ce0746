"""Groth16 verification (mirrors zkp/groth16/verifying.py:17-40).

e(A, B) == e(alpha, beta) * e(sum_pub r_i * sigma1_3[i], gamma) * e(C, delta).  The public-input
combination is a (small) G1 MSM on the GPU backend; the pairings run on the host
(zk_pairing / zk_pairing_check in libzkhip, see csrc/pairing.hip)."""
from ..field import ec_neg, ec_pairing, msm_g1, pairing_check


def lhs(prf_A, prf_B):
    """verifying.py:17-18"""
    return ec_pairing(prf_B, prf_A)


def _pub_combination(sigma1_3, rx_pub):
    idx = [i for i, _ in rx_pub]
    return msm_g1([ri for _, ri in rx_pub], [sigma1_3[i] for i in idx])


def rhs(prf_C, sigma1_1, sigma1_3, sigma2_1, rx_pub):
    """verifying.py:20-26"""
    temp = _pub_combination(sigma1_3, rx_pub)
    return (ec_pairing(sigma2_1[0], sigma1_1[0]) * ec_pairing(sigma2_1[1], temp)) * ec_pairing(sigma2_1[2], prf_C)


def verify(prf_A, prf_B, prf_C, sigma1_1, sigma1_3, sigma2_1, rx_pub):
    """verifying.py:29-40: LHS == RHS, evaluated as ONE product of four Miller loops with a shared final
    exponentiation:  e(A,B) * e(-alpha,beta) * e(-temp,gamma) * e(-C,delta) == 1  (same predicate)."""
    temp = _pub_combination(sigma1_3, rx_pub)
    return pairing_check([(prf_A, prf_B), (ec_neg(sigma1_1[0]), sigma2_1[0]),
                          (ec_neg(temp), sigma2_1[1]), (ec_neg(prf_C), sigma2_1[2])])
