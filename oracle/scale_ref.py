"""oracle/scale_ref.py -- TEST INFRASTRUCTURE ONLY (imported by tests/ and by bench.py / tools/bench_groth16.py as the CHECKER of
`verified_closed_form`; never by the product path).

Expected values for Groth16 at scale (BASELINE.json configs[3]).  The reference cannot run 2^20 constraints (dense W x G
matrices, O(W^2) hxr: zkp/groth16/poly_utils.py:116-125), but its own completeness check does not depend on the size: with the
toxic waste known, proof_A == A*G1, proof_B == B*G2, proof_C == C*G1 for scalars A, B, C in F_r
(zkp/groth16/test.py:303-325; general public indices zkp/groth16/arb_private/test.py:359-394).  This module evaluates those
scalars for the synthetic chain R1CS from its DEFINITION -- row k: A = B = e_{1+k}, C = e_{2+k} - e_{1+k} - c_k e_0; wires
[one, t_0 .. t_m]; public wires [0, 1]; QAP over the roots-of-unity domain -- with the C oracle's inverse NTT + Horner and
Python integers.  Nothing of zkhip's CRS, field layer or kernels is read.  Checked against the slow definition (explicit
Lagrange products, wire-by-wire sums) in tests/test_scale_expectations.py."""
import c_oracle as co
import py_ref as pr


def chain_witness(consts, t0):
    m = len(consts)
    t = [0] * (m + 1)
    t[0] = t0
    for k in range(m):
        t[k + 1] = (t[k] * t[k] + t[k] + consts[k]) % pr.R
    return [1] + t


def lagrange_at(m, k, x):
    """L_k(x) over H = {w^j}: (x^m - 1) / m * w^k / (x - w^k), by the formula, one index at a time."""
    R = pr.R
    wk = pow(pr.get_root_of_unity(m), k, R)
    return (pow(x, m, R) - 1) * pow(m, -1, R) % R * wk % R * pow((x - wk) % R, -1, R) % R


def chain_closed_form_oracle(consts, w, toxic, r, s):
    """(A, B, C) in F_r with proof_A = A*G1, proof_B = B*G2, proof_C = C*G1 (zkp/groth16/test.py:303-325):
    A = alpha + sum_i w_i A_i(x) + r delta, B likewise with s, C = (sum_{i not public} w_i (beta A_i + alpha B_i + C_i)(x)
    + H(x) Z(x)) / delta + s A + r B - r s delta.  The polynomial values come from the oracle's inverse NTT + Horner."""
    R = pr.R
    m = len(consts)
    al, be, de, x = (toxic[k] % R for k in ("alpha", "beta", "delta", "x"))
    t = w[1:]
    omega = pr.get_root_of_unity(m)
    at_x = lambda evals: co.fr_horner_arr(co.ntt_arr(co.to_limbs(evals), omega, inverse=True), x)
    a_x = at_x(t[:m])                                                  # sum_i w_i A_i(x): the row values interpolated over H
    b_x = a_x
    c_x = at_x([(t[k + 1] - t[k] - consts[k]) % R for k in range(m)])
    z_x = (pow(x, m, R) - 1) % R
    h_x = (a_x * b_x - c_x) * pow(z_x, -1, R) % R
    # public wires 0 (one) and 1 (t_0):  A_0 = B_0 = 0, C_0 = -sum_k c_k L_k;  A_1 = B_1 = L_0, C_1 = -L_0
    L0 = lagrange_at(m, 0, x)
    pub_a = t[0] * L0 % R
    pub_c = (-at_x([c % R for c in consts]) - t[0] * L0) % R
    dinv = pow(de, -1, R)
    priv = (be * (a_x - pub_a) + al * (b_x - pub_a) + (c_x - pub_c)) % R * dinv % R
    A = (al + a_x + r * de) % R
    B = (be + b_x + s * de) % R
    C = (priv + h_x * z_x % R * dinv + A * s + B * r - r * s % R * de) % R
    return A, B, C


def chain_crs_scalars(consts, toxic, idx_12, idx_14, idx_15):
    """Discrete logarithms of a few CRS elements of the chain circuit (zkp/groth16/setup.py:18-69 with the roots-of-unity QAP):
    sigma1_2[j] = x^j, sigma1_4[i] = (beta A_i + alpha B_i + C_i)(x) / delta for private wires i >= 2, sigma1_5[k] = x^k Z(x) / delta."""
    R = pr.R
    m = len(consts)
    al, be, de, x = (toxic[k] % R for k in ("alpha", "beta", "delta", "x"))
    dinv = pow(de, -1, R)
    z_x = (pow(x, m, R) - 1) % R
    s12 = [pow(x, j, R) for j in idx_12]
    s14 = []
    for i in idx_14:
        k = i - 1                                                      # wire i = t_k: in A and B of row k (k < m), +1 in C of row k-1, -1 in C of row k
        lk = lagrange_at(m, k, x) if k < m else 0
        lk1 = lagrange_at(m, k - 1, x)
        s14.append(((be + al) * lk + lk1 - lk) % R * dinv % R)
    s15 = [pow(x, k, R) * z_x % R * dinv % R for k in idx_15]
    return s12, s14, s15
