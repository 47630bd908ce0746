"""Groth16 CRS ("sigma") generation on the GPU backend (mirrors zkp/groth16/setup.py:15-69).

Every list is one fixed-base batch: the scalars are formed on the host in F_r exactly as the
reference forms them, then all `k_i * G` go to the GPU in a single zk_fixed_base_g1/g2 call
instead of one bn128.multiply per element.
"""
from ..field import FQ, FR, G1, G2, fixed_base_mul

g1 = G1
g2 = G2


def sigma11(alpha, beta, delta):
    """[alpha*G1, beta*G1, delta*G1]  (setup.py:15-16)"""
    return fixed_base_mul(g1, [int(alpha), int(beta), int(delta)])


def sigma12(numGates, x_val):
    """[x^i * G1 for i < numGates]  (setup.py:18-23)"""
    x_val = FR(x_val)
    return fixed_base_mul(g1, [int(x_val ** i) for i in range(numGates)])


def _abc_over(numWires, alpha, beta, div, Ax_val, Bx_val, Cx_val, keep):
    idx = [i for i in range(numWires) if keep(i)]
    vals = [(beta * Ax_val[i] + alpha * Bx_val[i] + Cx_val[i]) / div for i in idx]
    pts = fixed_base_mul(g1, [int(v) for v in vals])
    return idx, vals, pts


def sigma13(numWires, alpha, beta, gamma, Ax_val, Bx_val, Cx_val, pub_r_indexs=None):
    """Public-wire query and VAL vector (setup.py:25-40).  Entries at private indices are the
    reference's (FQ(0), FQ(0)) placeholders -- not curve points, never added."""
    if pub_r_indexs is None:
        pub_r_indexs = [0, 1]
    alpha, beta, gamma = FR(alpha), FR(beta), FR(gamma)
    idx, vals, pts = _abc_over(numWires, alpha, beta, gamma, Ax_val, Bx_val, Cx_val, lambda i: i in pub_r_indexs)
    sigma1_3 = [(FQ(0), FQ(0))] * numWires
    VAL = [FR(0)] * numWires
    for i, v, p in zip(idx, vals, pts):
        sigma1_3[i] = p
        VAL[i] = v
    return sigma1_3, VAL


def sigma14(numWires, alpha, beta, delta, Ax_val, Bx_val, Cx_val, pub_r_indexs=None):
    """Private-wire (L) query (setup.py:42-54); placeholders at the public indices."""
    if pub_r_indexs is None:
        pub_r_indexs = [0, 1]
    alpha, beta, delta = FR(alpha), FR(beta), FR(delta)
    idx, _, pts = _abc_over(numWires, alpha, beta, delta, Ax_val, Bx_val, Cx_val, lambda i: i not in pub_r_indexs)
    sigma1_4 = [(FQ(0), FQ(0))] * numWires
    for i, p in zip(idx, pts):
        sigma1_4[i] = p
    return sigma1_4


def sigma15(numGates, delta, x_val, Zx_val):
    """H query [(x^i * Z(x) / delta) * G1 for i < numGates-1]  (setup.py:56-60)"""
    x_val, delta, Zx_val = FR(x_val), FR(delta), FR(Zx_val)
    return fixed_base_mul(g1, [int((x_val ** i * Zx_val) / delta) for i in range(numGates - 1)])


def sigma21(beta, delta, gamma):
    """[beta*G2, gamma*G2, delta*G2]  (setup.py:62-63)"""
    return fixed_base_mul(g2, [int(beta), int(gamma), int(delta)])


def sigma22(numGates, x_val):
    """[x^i * G2 for i < numGates]  (setup.py:65-69)"""
    x_val = FR(x_val)
    return fixed_base_mul(g2, [int(x_val ** i) for i in range(numGates)])
