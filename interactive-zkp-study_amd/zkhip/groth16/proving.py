"""Groth16 proof elements on the GPU backend (mirrors zkp/groth16/proving.py:23-81).

The reference evaluates  sum_i Rx[i] * (sum_j M[i][j] * sigma[j])  as W*G scalar multiplications
inside the group.  That expression is linear, so here the scalars are collapsed on the host,
u_j = sum_i Rx[i] * M[i][j] in F_r (W*G field mul-adds), and each proof element becomes ONE
multi-scalar multiplication over [sigma1_1[k], sigma1_2[0..G), ...] on the GPU.  The group
element is identical (group law is exact; outputs are canonical affine).
"""
from ..field import FR, CURVE_ORDER, msm_g1, msm_g2
from .poly_utils import getNumGates, getNumWires

pointInf1 = None
pointInf2 = None


def _collapse(M, Rx):
    """u_j = sum_i Rx[i] * M[i][j] mod r."""
    numWires, numGates = getNumWires(M), getNumGates(M)
    r = [int(v) for v in Rx]
    u = [0] * numGates
    for i in range(numWires):
        ri = r[i]
        if ri == 0:
            continue
        row = M[i]
        for j in range(numGates):
            u[j] += ri * int(row[j])
    return [v % CURVE_ORDER for v in u]


def proof_a(sigma1_1, sigma1_2, Ax, Rx, r):
    """alpha*G1 + sum_i Rx[i]*A_i(x)*G1 + r*delta*G1   (proving.py:23-33)"""
    numGates = getNumGates(Ax)
    scalars = [1] + _collapse(Ax, Rx) + [int(r)]
    points = [sigma1_1[0]] + list(sigma1_2[:numGates]) + [sigma1_1[2]]
    return msm_g1(scalars, points)


def proof_b(sigma2_1, sigma2_2, Bx, Rx, s):
    """beta*G2 + sum_i Rx[i]*B_i(x)*G2 + s*delta*G2   (proving.py:35-45)"""
    numGates = getNumGates(Bx)
    scalars = [1] + _collapse(Bx, Rx) + [int(s)]
    points = [sigma2_1[0]] + list(sigma2_2[:numGates]) + [sigma2_1[2]]
    return msm_g2(scalars, points)


def proof_c(sigma1_1, sigma1_2, sigma1_4, sigma1_5, Bx, Rx, Hx, s, r, prf_A, pub_r_indexs=None):
    """s*A + r*B1 - r*s*delta*G1 + sum_{i not pub} Rx[i]*sigma1_4[i] + sum_{i<G-1} Hx[i]*sigma1_5[i]
    with B1 = beta*G1 + sum_j u_j*sigma1_2[j] + s*delta*G1   (proving.py:47-75).

    Everything is linear in the CRS points and prf_A, so it is one G1 MSM:
      r*B1 - r*s*delta*G1 = r*beta*G1 + sum_j (r*u_j)*sigma1_2[j]   (the +r*s*delta and -r*s*delta cancel)
    """
    if pub_r_indexs is None:
        pub_r_indexs = [0, 1]
    numGates, numWires = getNumGates(Bx), getNumWires(Bx)
    s, r = int(s) % CURVE_ORDER, int(r) % CURVE_ORDER
    u = _collapse(Bx, Rx)
    scalars = [s, r]
    points = [prf_A, sigma1_1[1]]
    scalars += [(r * uj) % CURVE_ORDER for uj in u]
    points += list(sigma1_2[:numGates])
    for i in range(numWires):
        if i in pub_r_indexs:
            continue  # placeholders (FQ(0), FQ(0)) at public indices are skipped (proving.py:66-70)
        scalars.append(int(Rx[i]))
        points.append(sigma1_4[i])
    for i in range(numGates - 1):
        scalars.append(int(Hx[i]))
        points.append(sigma1_5[i])
    return msm_g1(scalars, points)


def build_rpub_enum(pub_r_indexs, r_vec):
    """proving.py:77-81"""
    return [(i, r_vec[i]) for i in pub_r_indexs]
