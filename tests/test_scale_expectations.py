"""CPU test of the oracle-side expectations the -m gpu Groth16-at-scale tests and bench.py compare against
(oracle/scale_ref.py: r1cs_closed_form, r1cs_crs_scalars): at small sizes they must equal the definition evaluated the
slow way -- Lagrange basis polynomials as explicit products, the QAP sums wire by wire over the DENSE matrices
(zkp/groth16/test.py:303-325, zkp/groth16/setup.py:42-60) -- for both synthetic circuits; and the circuits' CSR data must
describe an R1CS their witness satisfies."""
import numpy as np
import pytest

import c_oracle as co
import py_ref as pr
from helpers import lagrange_at, r1cs_closed_form, r1cs_crs_scalars
from zkhip.groth16.circuits import BoolChainCircuit, ChainCircuit

R = pr.R
TOXIC = dict(alpha=3926, beta=3604, gamma=2971, delta=1357, x=3721 + (1 << 200))


def _lagrange_products(m, x):
    w = pr.get_root_of_unity(m)
    pts = [pow(w, j, R) for j in range(m)]
    out = []
    for k in range(m):
        num = den = 1
        for j in range(m):
            if j != k:
                num = num * (x - pts[j]) % R
                den = den * (pts[k] - pts[j]) % R
        out.append(num * pow(den, -1, R) % R)
    return out


def _dense(csr, m, W):
    row_ptr, col, vals = csr
    M = [[0] * W for _ in range(m)]
    v = co.from_limbs(vals)
    for k in range(m):
        for e in range(int(row_ptr[k]), int(row_ptr[k + 1])):
            M[k][int(col[e])] = (M[k][int(col[e])] + v[e]) % R
    return M


@pytest.mark.parametrize("kind,log_m", [("chain", 1), ("chain", 3), ("chain", 5), ("bool", 1), ("bool", 3), ("bool", 5)])
def test_expectations_equal_the_definition(kind, log_m):
    circ = (ChainCircuit if kind == "chain" else BoolChainCircuit)(log_m, seed=11)
    m, W = circ.m, circ.num_wires
    w, a, b, c = circ.witness()
    csr = circ.r1cs_csr()
    A, B, C = (_dense(csr[k], m, W) for k in "ABC")
    dot = lambda row: sum(x * y for x, y in zip(row, w)) % R
    assert [dot(r) for r in A] == a and [dot(r) for r in B] == b and [dot(r) for r in C] == c
    assert all(a[k] * b[k] % R == c[k] for k in range(m))                # the witness satisfies the R1CS
    if kind == "bool":
        assert sum(1 for v in w if v in (0, 1)) * 2 >= W                 # at least half of the wires are bits
    x, al, be, de = TOXIC["x"], TOXIC["alpha"], TOXIC["beta"], TOXIC["delta"]
    L = _lagrange_products(m, x)
    assert [lagrange_at(m, k, x) for k in range(m)] == L
    col_at = lambda M, i: sum(M[k][i] * L[k] for k in range(m)) % R      # M_i(x)
    Ai, Bi, Ci = ([col_at(M, i) for i in range(W)] for M in (A, B, C))
    zx = (pow(x, m, R) - 1) % R
    dinv = pow(de, -1, R)
    r, s = 4106, 4565
    a_x = sum(w[i] * Ai[i] for i in range(W)) % R
    b_x = sum(w[i] * Bi[i] for i in range(W)) % R
    c_x = sum(w[i] * Ci[i] for i in range(W)) % R
    h_x = (a_x * b_x - c_x) * pow(zx, -1, R) % R
    sA = (al + a_x + r * de) % R
    sB = (be + b_x + s * de) % R
    priv = sum(w[i] * ((be * Ai[i] + al * Bi[i] + Ci[i]) % R) for i in range(W) if i not in circ.pub) % R * dinv % R
    sC = (priv + h_x * zx % R * dinv + s * sA + r * sB - r * s * de) % R
    assert r1cs_closed_form(csr, w, circ.pub, TOXIC, r, s) == (sA, sB, sC)
    i14 = [i for i in range(W) if i not in circ.pub]
    s12, s14, s15 = r1cs_crs_scalars(csr, TOXIC, [0, m - 1], i14, [0, max(m - 2, 0)])
    assert s12 == [1, pow(x, m - 1, R)]
    assert s14 == [(be * Ai[i] + al * Bi[i] + Ci[i]) % R * dinv % R for i in i14]
    assert s15 == [zx * dinv % R, pow(x, max(m - 2, 0), R) * zx % R * dinv % R]


def test_column_of_a_wire_in_many_rows_goes_through_the_transform():
    """Wire 0 (`one`) of the chain circuit's C matrix sits in every row: scale_ref.column_at then interpolates the dense column
    with the oracle's inverse NTT + Horner instead of summing Lagrange terms -- both must agree."""
    import scale_ref
    circ = ChainCircuit(7, seed=5)
    csr = circ.r1cs_csr()["C"]
    x = TOXIC["x"]
    direct = sum((-c) % R * lagrange_at(circ.m, k, x) for k, c in enumerate(circ.consts)) % R
    assert scale_ref.column_at(csr, 0, circ.m, x) == direct


def test_csr_limbs_are_canonical():
    for circ in (ChainCircuit(4, seed=1), BoolChainCircuit(4, seed=1)):
        for name, (row_ptr, col, vals) in circ.r1cs_csr().items():
            assert row_ptr.dtype == np.uint32 and col.dtype == np.uint32 and vals.dtype == np.uint64
            assert row_ptr[0] == 0 and row_ptr[-1] == col.shape[0] == vals.shape[0]
            assert all(0 <= v < R for v in co.from_limbs(vals)) and int(col.max()) < circ.num_wires


def test_row_subset_of_a_csr_matrix():
    """zkhip.groth16.prover_dist._rows_subset (the constraint rows a rank owns, in block-cyclic order) against dense indexing."""
    from zkhip.groth16.prover_dist import _rows_subset
    circ = BoolChainCircuit(5, seed=2)
    m, W = circ.m, circ.num_wires
    rows = (np.arange(8, 16)[:, None] + 16 * np.arange(2)[None, :]).reshape(-1) % m      # a BC-style pick, repeated rows allowed
    for name, csr in circ.r1cs_csr().items():
        sub = _rows_subset(csr, rows.astype(np.int64))
        full = _dense(csr, m, W)
        assert _dense(sub, rows.shape[0], W) == [full[int(k)] for k in rows], name


@pytest.mark.parametrize("split", [2, 4, 1024])
def test_run_pointers_cut_groups_into_bounded_runs(monkeypatch, split):
    """prover_ntt._run_pointers (host logic of the device key generation): groups of consecutive entries are cut into runs of at most
    SPLIT entries; runs of one group stay consecutive, cover it exactly, and empty groups get no run -- applied repeatedly (the
    levels of _transposed_times) every group ends up as one sum."""
    import torch
    from zkhip.groth16 import prover_ntt
    monkeypatch.setattr(prover_ntt, "SPLIT", split)
    rng = np.random.default_rng(split)
    lens = torch.from_numpy(rng.integers(0, 40, size=50).astype(np.int64))
    lens[7] = 0
    lens[11] = 1000
    vals = torch.arange(int(lens.sum()), dtype=torch.int64) % 97 + 1              # the "entries": sums are checked through all levels
    want = [int(v.sum()) for v in torch.split(vals, [int(x) for x in lens])]
    cur, groups, levels = vals, lens, 0
    while True:
        ptr, runs = prover_ntt._run_pointers(groups)
        assert int(ptr[0]) == 0 and int(ptr[-1]) == cur.shape[0] and bool((ptr[1:] >= ptr[:-1]).all())
        assert int((ptr[1:] - ptr[:-1]).max()) <= split and int(runs.sum()) == ptr.shape[0] - 1
        assert bool((runs == (groups + split - 1) // split).all())
        cur = torch.stack([cur[int(a):int(b)].sum() for a, b in zip(ptr[:-1], ptr[1:])]) if ptr.shape[0] > 1 else cur[:0]
        groups, levels = runs, levels + 1
        if int(groups.max()) <= 1:
            break
    got, pos = [], 0
    for g in groups:
        got.append(int(cur[pos]) if int(g) else 0)
        pos += int(g)
    assert got == want and levels >= (1 if split >= 1000 else 3)
