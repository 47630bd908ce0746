"""F_r polynomial helpers of the Groth16 flow (mirrors zkp/groth16/poly_utils.py).

These are the scalar-side helpers the reference's callers use around the MSMs.  At the toy
sizes of the reference's QAP (integer domain {1..k}, dense W x G matrices) they are O(W*G)
host glue on Python ints; the at-scale quotient H(x) uses the NTT path instead
(zkhip.groth16.prover_ntt).  Quirks of the reference are kept on purpose:
`_multiply_vec_matrix` returns a length-W vector and asserts W != G (poly_utils.py:52-59).
"""
from ..field import FR


def _multiply_polys(a, b):
    """Coefficient convolution (poly_utils.py:17-22)."""
    out = [0] * (len(a) + len(b) - 1)
    for shift, coeff in enumerate(a):
        for k, other in enumerate(b, start=shift):
            out[k] += coeff * other
    return out


def _add_polys(a, b, subtract=False):
    """Coefficient-wise a + b, or a - b with subtract=True (poly_utils.py:25-31); the shorter operand is zero-extended."""
    sign = -1 if subtract else 1
    longer = max(len(a), len(b))
    left = list(a) + [0] * (longer - len(a))
    right = list(b) + [0] * (longer - len(b))
    return [x + sign * y for x, y in zip(left, right)]


def _subtract_polys(a, b):
    return _add_polys(a, b, subtract=True)


def _div_polys(a, b):
    """Schoolbook long division a = q * b + rem (poly_utils.py:37-45) -> (q, rem); rem keeps len(b) - 1 coefficients.
    Works on the reference's element types (FR, or plain numbers with true division)."""
    rem = list(a)
    top = len(b) - 1
    quot = [0] * (len(a) - top)
    for pos in range(len(quot) - 1, -1, -1):
        factor = rem[pos + top] / b[top]
        quot[pos] = factor
        for k, coeff in enumerate(b):
            rem[pos + k] = rem[pos + k] - factor * coeff
        rem.pop()
    return quot, rem


def _eval_poly(poly, x):
    """poly(x) by Horner's rule (same value as the power sum of poly_utils.py:48-49)."""
    acc = 0
    for coeff in reversed(poly):
        acc = acc * x + coeff
    return acc


def _multiply_vec_matrix(vec, matrix):
    """vec . matrix (poly_utils.py:52-59).  Kept as the reference has it: the result carries len(vec) entries, of which
    only the first len(matrix[0]) are ever written, and square shapes are refused by the assertion."""
    columns = len(matrix[0])
    assert len(vec) != columns
    out = [FR(0)] * len(vec)
    for j, column in enumerate(zip(*matrix)):
        total = FR(0)
        for weight, entry in zip(vec, column):
            total = total + weight * entry
        out[j] = total
    return out


def _multiply_vec_vec(vec1, vec2):
    assert len(vec1) == len(vec2)
    total = 0
    for x, y in zip(vec1, vec2):
        total += x * y
    return total


def getNumWires(Ax):
    return len(Ax)


def getNumGates(Ax):
    return len(Ax[0])


def getFRPoly1D(poly):
    """Floats of the QAP front end -> FR by round(), not int() (poly_utils.py:75-76)."""
    return [FR(round(v)) for v in poly]


def getFRPoly2D(poly):
    return [getFRPoly1D(row) for row in poly]


def ax_val(Ax, x_val):
    """Every wire polynomial evaluated at x (poly_utils.py ax_val / bx_val / cx_val)."""
    return [_eval_poly(row, x_val) for row in Ax]


bx_val = ax_val
cx_val = ax_val


def zx_val(Zx, x_val):
    return _eval_poly(Zx, x_val)


def hx_val(Hx, x_val):
    return _eval_poly(Hx, x_val)


def hxr(Ax, Bx, Cx, Zx, R):
    """H and remainder of (R.A * R.B - R.C) / Z (poly_utils.py:116-125)."""
    u_a, u_b, u_c = (_multiply_vec_matrix(R, M) for M in (Ax, Bx, Cx))
    numerator = _subtract_polys(_multiply_polys(u_a, u_b), u_c)
    return _div_polys(numerator, Zx)
