"""F_r polynomial helpers of the Groth16 flow (mirrors zkp/groth16/poly_utils.py).

These are the scalar-side helpers the reference's callers use around the MSMs.  At the toy
sizes of the reference's QAP (integer domain {1..k}, dense W x G matrices) they are O(W*G)
host glue on Python ints; the at-scale quotient H(x) uses the NTT path instead
(zkhip.groth16.prover_ntt).  Quirks of the reference are kept on purpose:
`_multiply_vec_matrix` returns a length-W vector and asserts W != G (poly_utils.py:52-59).
"""
from ..field import FR


def _multiply_polys(a, b):
    """poly_utils.py:17-22"""
    o = [0] * (len(a) + len(b) - 1)
    for i in range(len(a)):
        for j in range(len(b)):
            o[i + j] += a[i] * b[j]
    return o


def _add_polys(a, b, subtract=False):
    """poly_utils.py:25-31"""
    o = [0] * max(len(a), len(b))
    for i in range(len(a)):
        o[i] += a[i]
    for i in range(len(b)):
        o[i] += b[i] * (-1 if subtract else 1)
    return o


def _subtract_polys(a, b):
    return _add_polys(a, b, subtract=True)


def _div_polys(a, b):
    """poly_utils.py:37-45: long division, returns (quotient, remainder)."""
    o = [0] * (len(a) - len(b) + 1)
    remainder = a
    while len(remainder) >= len(b):
        leading_fac = remainder[-1] / b[-1]
        pos = len(remainder) - len(b)
        o[pos] = leading_fac
        remainder = _subtract_polys(remainder, _multiply_polys(b, [0] * pos + [leading_fac]))[:-1]
    return o, remainder


def _eval_poly(poly, x):
    """poly_utils.py:48-49"""
    return sum([poly[i] * x ** i for i in range(len(poly))])


def _multiply_vec_matrix(vec, matrix):
    """poly_utils.py:52-59 (result has len(vec) entries; asserts W != G like the reference)."""
    assert not len(vec) == len(matrix[0])
    target = [FR(0)] * len(vec)
    for i in range(len(matrix)):
        for j in range(len(matrix[0])):
            target[j] = target[j] + vec[i] * matrix[i][j]
    return target


def _multiply_vec_vec(vec1, vec2):
    assert len(vec1) == len(vec2)
    target = 0
    for i in range(len(vec1)):
        target += vec1[i] * vec2[i]
    return target


def getNumWires(Ax):
    return len(Ax)


def getNumGates(Ax):
    return len(Ax[0])


def getFRPoly1D(poly):
    """poly_utils.py:75-76 (round(), not int())."""
    return [FR(round(num)) for num in poly]


def getFRPoly2D(poly):
    return [[FR(round(num)) for num in vec] for vec in poly]


def ax_val(Ax, x_val):
    return [_eval_poly(p, x_val) for p in Ax]


bx_val = ax_val
cx_val = ax_val


def zx_val(Zx, x_val):
    return _eval_poly(Zx, x_val)


def hx_val(Hx, x_val):
    return _eval_poly(Hx, x_val)


def hxr(Ax, Bx, Cx, Zx, R):
    """(R.A * R.B - R.C) / Z -> (H, remainder)   (poly_utils.py:116-125)."""
    Rax = _multiply_vec_matrix(R, Ax)
    Rbx = _multiply_vec_matrix(R, Bx)
    Rcx = _multiply_vec_matrix(R, Cx)
    Px = _subtract_polys(_multiply_polys(Rax, Rbx), Rcx)
    q, r = _div_polys(Px, Zx)
    return q, r
