"""KZG commitment on the GPU backend (mirrors zkp/plonk/kzg.py:32-67)."""
from ..field import FR, G1, ec_add, ec_mul, ec_neg, msm_g1, pairing_check
from .polynomial import Polynomial, poly_div


def commit(poly, srs):
    """C = sum_i c_i * [tau^i]_1 as one G1 MSM (kzg.py:32-67).

    Raises ValueError when the degree exceeds the SRS (kzg.py:54-57); the zero polynomial
    commits to None (kzg.py:60, tests/plonk/test_crypto.py:132-136).  Zero coefficients
    contribute nothing (kzg.py:62-63) -- the MSM skips zero digits by construction."""
    if poly.degree > srs.max_degree:
        raise ValueError("polynomial degree %d exceeds the SRS max degree %d" % (poly.degree, srs.max_degree))
    coeffs = poly.coeffs
    return msm_g1(coeffs, srs.g1_powers[:len(coeffs)])


def create_witness(poly, point, srs):
    """Opening proof pi = commit((p(x) - p(z)) / (x - z))  (kzg.py:70-114)."""
    if not isinstance(point, FR):
        point = FR(point)
    y = poly.evaluate(point)
    quotient, remainder = poly_div(poly - Polynomial([y]), Polynomial([FR(0) - point, FR(1)]))
    for c in remainder.coeffs:
        if c != FR(0):
            raise ValueError("opening proof: non-zero remainder")
    return commit(quotient, srs)


def verify_opening(commitment, proof, point, evaluation, srs):
    """e(C - y*G1, G2) == e(pi, [tau - z]_2)  (kzg.py:117-160), checked as one pairing product
    e(C - y*G1, G2) * e(-pi, [tau - z]_2) == 1."""
    if not isinstance(point, FR):
        point = FR(point)
    if not isinstance(evaluation, FR):
        evaluation = FR(evaluation)
    tau_minus_z_g2 = ec_add(srs.g2_powers[1], ec_neg(ec_mul(srs.g2_powers[0], point)))
    c_minus_y = ec_add(commitment, ec_neg(ec_mul(G1, evaluation)))
    return pairing_check([(c_minus_y, srs.g2_powers[0]), (ec_neg(proof), tau_minus_z_g2)])
