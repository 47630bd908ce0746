"""KZG commitment on the GPU backend (mirrors zkp/plonk/kzg.py:32-67)."""
from ..field import msm_g1


def commit(poly, srs):
    """C = sum_i c_i * [tau^i]_1 as one G1 MSM (kzg.py:32-67).

    Raises ValueError when the degree exceeds the SRS (kzg.py:54-57); the zero polynomial
    commits to None (kzg.py:60, tests/plonk/test_crypto.py:132-136).  Zero coefficients
    contribute nothing (kzg.py:62-63) -- the MSM skips zero digits by construction."""
    if poly.degree > srs.max_degree:
        raise ValueError("polynomial degree %d exceeds the SRS max degree %d" % (poly.degree, srs.max_degree))
    coeffs = poly.coeffs
    return msm_g1(coeffs, srs.g1_powers[:len(coeffs)])
