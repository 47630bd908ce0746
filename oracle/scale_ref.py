"""oracle/scale_ref.py -- TEST INFRASTRUCTURE ONLY (imported by tests/ and by bench.py / tools/bench_groth16.py as the CHECKER of
`verified_closed_form`; never by the product path).

Expected values for Groth16 at scale (BASELINE.json configs[3]).  The reference cannot run 2^20 constraints (dense W x G
matrices, O(W^2) hxr: zkp/groth16/poly_utils.py:116-125), but its own completeness check does not depend on the size: with the
toxic waste known, proof_A == A*G1, proof_B == B*G2, proof_C == C*G1 for scalars A, B, C in F_r
(zkp/groth16/test.py:303-325; general public indices zkp/groth16/arb_private/test.py:359-394).  This module evaluates those
scalars, and the discrete logarithms of CRS elements (zkp/groth16/setup.py:18-69), for an R1CS given as DATA -- three CSR
matrices (row_ptr uint32[m+1], col uint32[nnz], vals (nnz, 4) uint64 canonical limbs), a witness, the public wire indices -- over
the roots-of-unity domain H = {w^k}, with the C oracle's sparse mat-vec, inverse NTT and Horner rule and Python integers.
Nothing of zkhip's CRS, field layer or kernels is read.  Checked against the slow definition (explicit Lagrange products,
wire-by-wire sums) in tests/test_scale_expectations.py."""
import numpy as np

import c_oracle as co
import py_ref as pr

R = pr.R


def lagrange_at(m, k, x):
    """L_k(x) over H = {w^j}: (x^m - 1) / m * w^k / (x - w^k), by the formula, one index at a time."""
    wk = pow(pr.get_root_of_unity(m), k, R)
    return (pow(x, m, R) - 1) * pow(m, -1, R) % R * wk % R * pow((x - wk) % R, -1, R) % R


def _interp_at(evals_limbs, x):
    """p(x) for the polynomial of degree < m with p(w^k) = evals[k]: the oracle's inverse NTT, then Horner."""
    m = evals_limbs.shape[0]
    return co.fr_horner_arr(co.ntt_arr(evals_limbs, pr.get_root_of_unity(m), inverse=True), x)


def column_at(csr, i, m, x):
    """M_i(x) = sum_k M[k][i] L_k(x) for wire i of one R1CS matrix (the QAP polynomial of the wire at x)."""
    row_ptr, col, vals = csr
    hits = np.nonzero(col == np.uint32(i))[0]
    if hits.shape[0] == 0:
        return 0
    rows = np.searchsorted(row_ptr, hits, side="right") - 1
    coef = co.from_limbs(vals[hits])
    if hits.shape[0] <= 64:
        return sum(c * lagrange_at(m, int(k), x) for c, k in zip(coef, rows)) % R
    assert np.unique(rows).shape[0] == rows.shape[0], "a wire twice in one row: merge the entries first"
    dense = np.zeros((m, 4), dtype=np.uint64)
    dense[rows] = vals[hits]
    return _interp_at(dense, x)


def r1cs_closed_form(csr, w, pub, toxic, r, s):
    """(A, B, C) in F_r with proof_A = A*G1, proof_B = B*G2, proof_C = C*G1 (zkp/groth16/test.py:303-325):
       A = alpha + sum_i w_i A_i(x) + r delta,   B = beta + sum_i w_i B_i(x) + s delta,
       C = (sum_{i not public} w_i (beta A_i + alpha B_i + C_i)(x) + H(x) Z(x)) / delta + s A + r B - r s delta,
    with H = (A.B - C) / Z on the roots-of-unity domain.  csr: {"A": .., "B": .., "C": ..}; w: list of ints."""
    m = csr["A"][0].shape[0] - 1
    al, be, de, x = (toxic[k] % R for k in ("alpha", "beta", "delta", "x"))
    W = co.to_limbs(w)
    at_x = {}
    for name in "ABC":
        evals = co.fr_spmv_arr(*csr[name], W)                            # the per-constraint values M.w
        at_x[name] = _interp_at(evals, x)                                # sum_i w_i M_i(x)
    a_x, b_x, c_x = at_x["A"], at_x["B"], at_x["C"]
    z_x = (pow(x, m, R) - 1) % R
    h_x = (a_x * b_x - c_x) * pow(z_x, -1, R) % R
    pub_part = {name: sum(w[i] * column_at(csr[name], i, m, x) for i in pub) % R for name in "ABC"}
    dinv = pow(de, -1, R)
    priv = (be * (a_x - pub_part["A"]) + al * (b_x - pub_part["B"]) + (c_x - pub_part["C"])) % R * dinv % R
    A = (al + a_x + r * de) % R
    B = (be + b_x + s * de) % R
    C = (priv + h_x * z_x % R * dinv + A * s + B * r - r * s % R * de) % R
    return A, B, C


def r1cs_crs_scalars(csr, toxic, idx_12, idx_14, idx_15):
    """Discrete logarithms of CRS elements (zkp/groth16/setup.py:18-69 over the roots-of-unity QAP):
    sigma1_2[j] = sigma2_2[j] = x^j, sigma1_4[i] = (beta A_i + alpha B_i + C_i)(x) / delta (private wires i),
    sigma1_5[k] = x^k Z(x) / delta."""
    m = csr["A"][0].shape[0] - 1
    al, be, de, x = (toxic[k] % R for k in ("alpha", "beta", "delta", "x"))
    dinv = pow(de, -1, R)
    z_x = (pow(x, m, R) - 1) % R
    s12 = [pow(x, j, R) for j in idx_12]
    s14 = [(be * column_at(csr["A"], i, m, x) + al * column_at(csr["B"], i, m, x) + column_at(csr["C"], i, m, x)) % R * dinv % R for i in idx_14]
    s15 = [pow(x, k, R) * z_x % R * dinv % R for k in idx_15]
    return s12, s14, s15
