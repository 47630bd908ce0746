"""Synthetic sparse R1CS instances for Groth16 at scale (BASELINE.json configs[3]: "synthetic 2^20-constraint R1CS").

The reference builds its R1CS from a toy Python program (zkp/groth16/code_to_r1cs.py) as dense W x G lists; at 2^20
constraints the matrices are given in CSR form (<= 3 non-zeros per row) as numpy arrays, produced without per-element
Python work so that key generation is not dominated by input marshalling:

    r1cs_csr()  -> {"A" | "B" | "C": (row_ptr uint32[m+1], col uint32[nnz], vals (nnz, 4) uint64 canonical limbs)}
    witness()   -> (w, a, b, c): the wire values and the per-constraint products A.w, B.w, C.w as Python ints
                   (an input generator: a serial recurrence, one modular product per constraint)

Public wires are [0, 1], the reference's default `pub_r_indexs` (zkp/groth16/proving.py:47-49)."""
import numpy as np

from ..field import CURVE_ORDER as R

_R_LIMBS = np.array([(R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def _small_limbs(v):
    """(n,) non-negative integers below 2^63 -> (n, 4) limbs."""
    out = np.zeros((v.shape[0], 4), dtype=np.uint64)
    out[:, 0] = v.astype(np.uint64)
    return out


def _neg_small_limbs(v):
    """(n,) integers 0 <= v < 2^60 -> limbs of (-v) mod r (r's low limb exceeds 2^60, so only it changes; -0 = 0)."""
    out = np.tile(_R_LIMBS, (v.shape[0], 1))
    out[:, 0] -= v.astype(np.uint64)
    out[v == 0] = 0
    return out


class ChainCircuit:
    """m = 2^log_m constraints t_{k+1} = t_k * t_k + t_k + c_k.

    Wires: 0 = one, 1 + k = t_k (k = 0..m).  Row k:  A = t_k,  B = t_k,  C = t_{k+1} - t_k - c_k * one.
    Every wire value is a full-size field element: the witness is uniform."""

    def __init__(self, log_m, seed=1):
        self.log_m = log_m
        self.m = 1 << log_m
        rng = np.random.default_rng(seed)
        self._consts = rng.integers(1, 1 << 30, size=self.m)
        self.t0 = int(rng.integers(2, 1 << 62))
        self.num_wires = self.m + 2
        self.pub = [0, 1]

    @property
    def consts(self):
        return [int(v) for v in self._consts]

    def witness(self):
        m, cs = self.m, self.consts
        t = [0] * (m + 1)
        t[0] = self.t0
        for k in range(m):
            t[k + 1] = (t[k] * t[k] + t[k] + cs[k]) % R
        w = [1] + t
        a = t[:m]
        c = [(t[k + 1] - t[k] - cs[k]) % R for k in range(m)]
        return w, a, list(a), c

    def r1cs_csr(self):
        m = self.m
        k = np.arange(m, dtype=np.uint32)
        one = _small_limbs(np.ones(m, dtype=np.uint64))
        ab = (np.arange(m + 1, dtype=np.uint32), 1 + k, one)
        col_c = np.stack([2 + k, 1 + k, np.zeros(m, dtype=np.uint32)], axis=1).reshape(-1)
        minus_one = np.tile(_neg_small_limbs(np.ones(1, dtype=np.uint64)), (m, 1))
        vals_c = np.stack([one, minus_one, _neg_small_limbs(self._consts)], axis=1).reshape(-1, 4)
        return {"A": ab, "B": ab, "C": (3 * np.arange(m + 1, dtype=np.uint32), col_c.astype(np.uint32), vals_c)}


class BoolChainCircuit:
    """A witness of the kind real circuits produce: half of the wires are bits (SURVEY.md section 7 'hard parts': "witness
    scalars are highly non-uniform (many 0/1)").  m = 2^log_m constraints (log_m >= 1), in pairs j = 0 .. m/2 - 1:

        row 2j      b_j * b_j = b_j                              (booleanity)
        row 2j + 1  t_j * (t_j + b_j) = t_{j+1} - c_j * one      (a chain step steered by the bit)

    Wires: 0 = one, 1 + j = t_j (j = 0..m/2), 2 + m/2 + j = b_j (j < m/2): m + 2 wires, m/2 + 1 of them in {0, 1}."""

    def __init__(self, log_m, seed=1):
        assert log_m >= 1
        self.log_m = log_m
        self.m = 1 << log_m
        self.h = self.m // 2
        rng = np.random.default_rng(seed)
        self._consts = rng.integers(1, 1 << 30, size=self.h)
        self._bits = rng.integers(0, 2, size=self.h)
        self.t0 = int(rng.integers(2, 1 << 62))
        self.num_wires = self.m + 2
        self.pub = [0, 1]

    @property
    def consts(self):
        return [int(v) for v in self._consts]

    def witness(self):
        h, cs, bs = self.h, self.consts, [int(v) for v in self._bits]
        t = [0] * (h + 1)
        t[0] = self.t0
        a, b, c = [0] * self.m, [0] * self.m, [0] * self.m
        for j in range(h):
            t[j + 1] = (t[j] * (t[j] + bs[j]) + cs[j]) % R
            a[2 * j] = b[2 * j] = c[2 * j] = bs[j]
            a[2 * j + 1] = t[j]
            b[2 * j + 1] = (t[j] + bs[j]) % R
            c[2 * j + 1] = (t[j + 1] - cs[j]) % R
        return [1] + t + bs, a, b, c

    def r1cs_csr(self):
        m, h = self.m, self.h
        j = np.arange(h, dtype=np.uint32)
        t_j, t_next, b_j = 1 + j, 2 + j, np.uint32(2 + h) + j
        one = lambda n: _small_limbs(np.ones(n, dtype=np.uint64))
        # A: one entry per row (b_j | t_j)
        col_a = np.stack([b_j, t_j], axis=1).reshape(-1)
        A = (np.arange(m + 1, dtype=np.uint32), col_a.astype(np.uint32), one(m))
        # B: row 2j: b_j;  row 2j+1: t_j, b_j
        rp_b = np.zeros(m + 1, dtype=np.uint32)
        rp_b[1::2] = 3 * j + 1
        rp_b[2::2] = 3 * j + 3
        col_b = np.stack([b_j, t_j, b_j], axis=1).reshape(-1)
        B = (rp_b, col_b.astype(np.uint32), one(3 * h))
        # C: row 2j: b_j;  row 2j+1: t_{j+1}, -c_j * one
        col_c = np.stack([b_j, t_next, np.zeros(h, dtype=np.uint32)], axis=1).reshape(-1)
        vals_c = np.stack([one(h), one(h), _neg_small_limbs(self._consts)], axis=1).reshape(-1, 4)
        C = (rp_b.copy(), col_c.astype(np.uint32), vals_c)
        return {"A": A, "B": B, "C": C}
