"""Groth16 at scale on the GPU backend: roots-of-unity QAP, H(x) by NTT, proof elements by MSM.

The reference's Groth16 flow (zkp/groth16/{qap_creator_lcm,poly_utils,setup,proving}.py) works on
the integer domain {1..k} with float Lagrange interpolation, a dense W x G coefficient matrix and
an O(W^2) `hxr` (poly_utils.py:116-125) -- none of which can reach 2^20 constraints (SURVEY.md
section 7 "hard parts", section 8 row A7).  This module keeps the reference's CRS *shape* and proof
formulas (sigma1_1, sigma1_2 = powers of x in G1, sigma1_4 = per-wire L query, sigma1_5 = H query,
sigma2_1, sigma2_2; proof_a/b/c of proving.py:23-75) and swaps the polynomial side for the standard
at-scale one:

  * QAP over H = {w^k}: A_i(w^k) = A[k][i], so the coefficient vectors u_A = R.A, u_B, u_C that the
    reference builds with `_multiply_vec_matrix` are inverse NTTs of the per-constraint dot products;
  * H(x) = (A(x)B(x) - C(x)) / (x^m - 1): 3 iNTT + 3 coset NTT + pointwise quotient + 1 coset iNTT
    (7 transforms of size m) instead of schoolbook multiply + long division;
  * proof_A = alpha*G1 + MSM(u_A, sigma1_2) + r*delta*G1, and so on: one device MSM per query.

Everything stays resident in HBM (torch tensors used as plain device buffers); correctness at any
size is checked against closed-form scalars computed from the known toxic waste
(zkp/groth16/test.py:303-325: proof_A == A*G1, proof_B == B*G2, proof_C == C*G1).
"""
import numpy as np

from .. import _lib
from ..device import FrVec, MsmPlan, NttPlan, fr_quotient
from ..field import CURVE_ORDER as R, G1, G2, fixed_base_mul, g1_to_limbs, g2_to_limbs, msm_g1, limbs_to_g1

COSET_SHIFT = 5  # the reference's coset generator (zkp/plonk/utils.py:166-167)


def _batch_inverse(vals):
    """Montgomery's trick on Python ints mod r."""
    n = len(vals)
    pref = [1] * (n + 1)
    for i, v in enumerate(vals):
        pref[i + 1] = pref[i] * v % R
    inv = pow(pref[n], -1, R)
    out = [0] * n
    for i in range(n - 1, -1, -1):
        out[i] = pref[i] * inv % R
        inv = inv * vals[i] % R
    return out


class ChainCircuit:
    """Synthetic R1CS with m = 2^log_m constraints t_{k+1} = t_k * t_k + t_k + c_k.

    Wires: 0 = one, 1 + k = t_k (k = 0..m); public wires [0, 1] (the reference's default
    pub_r_indexs).  Row k:  A = t_k,  B = t_k,  C = t_{k+1} - t_k - c_k * one  (<= 3 non-zeros)."""

    def __init__(self, log_m, seed=1):
        self.log_m = log_m
        self.m = 1 << log_m
        rng = np.random.default_rng(seed)
        self.consts = [int(v) for v in rng.integers(1, 1 << 30, size=self.m)]
        self.t0 = int(rng.integers(2, 1 << 62))
        self.num_wires = self.m + 2
        self.pub = [0, 1]

    def witness(self):
        """-> (w, a_evals, b_evals, c_evals): the wire values and the per-constraint products."""
        t = [0] * (self.m + 1)
        t[0] = self.t0
        for k in range(self.m):
            t[k + 1] = (t[k] * t[k] + t[k] + self.consts[k]) % R
        w = [1] + t
        a = t[:self.m]
        c = [(t[k + 1] - t[k] - self.consts[k]) % R for k in range(self.m)]
        return w, a, list(a), c

    def r1cs_csr(self):
        """The R1CS matrices in CSR form, {name: (row_ptr u32[m+1], col u32[nnz], vals (nnz, 4) u64 limbs)}:
        row k:  A = B = e_{1+k};  C = e_{2+k} - e_{1+k} - c_k * e_0."""
        m = self.m
        k = np.arange(m, dtype=np.uint32)
        one = np.zeros((m, 4), dtype=np.uint64)
        one[:, 0] = 1
        ab = (np.arange(m + 1, dtype=np.uint32), 1 + k, one)
        col_c = np.stack([2 + k, 1 + k, np.zeros(m, dtype=np.uint32)], axis=1).reshape(-1)
        minus_one = _lib.ints_to_limbs([R - 1])[0]
        minus_c = _lib.ints_to_limbs([(-c) % R for c in self.consts])
        vals_c = np.stack([one, np.tile(minus_one, (m, 1)), minus_c], axis=1).reshape(-1, 4)
        return {"A": ab, "B": ab, "C": (3 * np.arange(m + 1, dtype=np.uint32), col_c.astype(np.uint32), vals_c)}

    def lagrange_at(self, x):
        """[L_k(x)] for the domain H = {w^k}: L_k(x) = (x^m - 1)/m * w^k / (x - w^k)."""
        m = self.m
        omega = pow(5, (R - 1) // m, R)
        roots, cur = [], 1
        for _ in range(m):
            roots.append(cur)
            cur = cur * omega % R
        zx = (pow(x, m, R) - 1) % R
        inv = _batch_inverse([(x - wk) % R for wk in roots])
        scale = zx * pow(m, -1, R) % R
        return [scale * roots[k] % R * inv[k] % R for k in range(m)], zx

    def qap_at(self, x):
        """Per-wire evaluations A_i(x), B_i(x), C_i(x) and Z(x) (lists of length num_wires)."""
        L, zx = self.lagrange_at(x)
        m, W = self.m, self.num_wires
        A = [0] * W
        C = [0] * W
        for k in range(m):
            A[1 + k] = L[k]
        C[0] = (-sum(self.consts[k] * L[k] for k in range(m))) % R
        for k in range(m):
            C[1 + k] = (C[1 + k] - L[k]) % R
            C[2 + k] = (C[2 + k] + L[k]) % R
        return A, list(A), C, zx


def _dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).cuda()


class ScaleCRS:
    """Device-resident CRS in the reference's sigma layout (zkp/groth16/setup.py:15-69)."""

    def __init__(self, circuit, alpha, beta, gamma, delta, x_val):
        self.circuit = circuit
        m, W = circuit.m, circuit.num_wires
        self.toxic = dict(alpha=alpha % R, beta=beta % R, gamma=gamma % R, delta=delta % R, x=x_val % R)
        Ax, Bx, Cx, zx = circuit.qap_at(x_val)
        self.Ax, self.Bx, self.Cx, self.Zx = Ax, Bx, Cx, zx
        dinv = pow(delta, -1, R)
        powers, cur = [], 1
        for _ in range(m):
            powers.append(cur)
            cur = cur * x_val % R
        # sigma1_1 / sigma2_1 (setup.py:15-16, 62-63)
        self.sigma1_1 = fixed_base_mul(G1, [alpha, beta, delta])
        self.sigma2_1 = fixed_base_mul(G2, [beta, gamma, delta])
        lib = _lib.load()
        S = _lib.ints_to_limbs(powers)
        g1 = g1_to_limbs([G1])
        g2 = g2_to_limbs([G2])
        # sigma1_2 = [x^j]_1, sigma2_2 = [x^j]_2  (setup.py:18-23, 65-69)
        s12 = np.zeros((m, 8), dtype=np.uint64)
        _lib.check(lib.zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(S), m, _lib.ptr(s12)))
        s22 = np.zeros((m, 16), dtype=np.uint64)
        _lib.check(lib.zk_fixed_base_g2(_lib.ptr(g2), _lib.ptr(S), m, _lib.ptr(s22)))
        # sigma1_4: L query for private wires; placeholders (zeros) at the public indices (setup.py:42-54)
        lq = [0 if i in circuit.pub else (beta * Ax[i] + alpha * Bx[i] + Cx[i]) % R * dinv % R for i in range(W)]
        self.l_scalars = lq
        s14 = np.zeros((W, 8), dtype=np.uint64)
        _lib.check(lib.zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(_lib.ints_to_limbs(lq)), W, _lib.ptr(s14)))
        for i in circuit.pub:
            s14[i] = 0
        # sigma1_5 = [x^k Z(x) / delta]_1, k < m - 1  (setup.py:56-60)
        hq = [powers[k] * zx % R * dinv % R for k in range(m - 1)]
        s15 = np.zeros((m - 1, 8), dtype=np.uint64)
        _lib.check(lib.zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(_lib.ints_to_limbs(hq)), m - 1, _lib.ptr(s15)))
        # query arrays with the constant terms appended, so that a proof element is ONE device MSM:
        #   G1: sigma1_2 | alpha*G1 | delta*G1 | beta*G1        G2: sigma2_2 | beta*G2 | delta*G2
        s12x = np.concatenate([s12, g1_to_limbs([self.sigma1_1[0], self.sigma1_1[2], self.sigma1_1[1]])])
        s22x = np.concatenate([s22, g2_to_limbs([self.sigma2_1[0], self.sigma2_1[2]])])
        self.d_s12, self.d_s22, self.d_s14, self.d_s15 = _dev(s12x), _dev(s22x), _dev(s14), _dev(s15)


class ScaleProver:
    """Plans + scratch for proving against one ScaleCRS."""

    def __init__(self, crs, chunk_log=0):
        """chunk_log: chunk size of the G1 plan's MSMs beyond it (0 = the library's 2^22); from 2^21 constraints on the merged
        proof_C query (3 m + 4 bases) runs as chunks in the lanes the A query leaves free."""
        import torch
        self.crs = crs
        c = crs.circuit
        self.m, self.W = c.m, c.num_wires
        self.ntt = NttPlan(c.log_m)
        # The CRS never changes: bind the three G1 queries as one array (sigma1_2+ | sigma1_4 | sigma1_5) and sigma2_2+ in G2:
        # tables of 2^(20 w) * P, 13 n bucket additions per MSM instead of 16 n (zk_msm_plan_bind_points).
        self.bound = self.m + 2 > (1 << 17)
        self.off14, self.off15 = self.m + 3, self.m + 3 + self.W
        self.n_c = self.off15 + self.m - 1                     # all three queries behind each other: (m+3) + W + (m-1) bases
        self.g1 = MsmPlan(_lib.GROUP_G1, self.n_c if self.bound else max(self.W, self.m + 3), chunk_log=chunk_log)
        self.g2 = MsmPlan(_lib.GROUP_G2, self.m + 2)
        if self.bound:
            st0 = torch.cuda.current_stream().cuda_stream
            all_g1 = torch.cat([crs.d_s12, crs.d_s14, crs.d_s15])
            self.g1.bind(all_g1.data_ptr(), all_g1.shape[0], st0)
            self.g2.bind(crs.d_s22.data_ptr(), self.m + 2, st0)
            del all_g1
        new = lambda rows: torch.empty((rows, 4), dtype=torch.int64, device="cuda")
        self.ext_a, self.ext_b1, self.ext_b2 = new(self.m + 3), new(self.m + 3), new(self.m + 2)
        self.scratch = [new(self.m) for _ in range(4)]
        # bound CRS: proof_C needs r*(beta + B(x)) + L + H only as a SUM, so the three queries run as ONE MSM over the whole bound
        # array with the scalars (r*u_B | 0 0 r | w | h) behind each other -- the same bucket additions, one sort and one bucket
        # reduction instead of three.  h (m coefficients, the last one zero) is computed in place at the tail of that buffer.
        self.sc_c = new(self.n_c + 1) if self.bound else None
        # the per-proof constant-term scalars (they depend on r and s) go up in ONE asynchronous copy from a pinned buffer
        self.h_consts = torch.empty((8, 4), dtype=torch.int64).pin_memory()
        self.d_consts = new(8)
        self.zinv = pow((pow(COSET_SHIFT, self.m, R) - 1) % R, -1, R)  # 1 / Z_H on the coset k*H

    def prove(self, d_a, d_b, d_c, d_w, r, s, stream=None):
        """d_a, d_b, d_c: device (m, 4) evaluations sum_i w_i A[k][i] etc. (d_c is overwritten with the
        coefficient vector u_C); d_w: device (W, 4) witness.
        -> (proof_A, proof_B, proof_C) as points, plus the device buffer of the H coefficients."""
        import torch
        if stream is not None and stream != torch.cuda.current_stream().cuda_stream:
            # the copies below are torch calls (current stream), the transforms and MSMs go to `stream`: run the whole proof with
            # `stream` as torch's current stream so that both are ordered on it
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                return self.prove(d_a, d_b, d_c, d_w, r, s, None)
        st = torch.cuda.current_stream().cuda_stream
        m, W, crs = self.m, self.W, self.crs
        r, s = r % R, s % R
        ca, cb, cc, h = self.scratch
        if self.bound:
            h = self.sc_c[self.off15:self.off15 + m]           # the H coefficients land where the merged MSM reads them
        ua, ub = self.ext_a[:m], self.ext_b1[:m]
        # constant-term scalars behind the coefficient vectors (see ScaleCRS; bases alpha | delta | beta resp. beta | delta):
        #   A: [1, r, 0]    B1: [0, 0, 1]    B2: [1, s]    merged proof_C query (bound CRS): [s, r*s, r]
        self.h_consts.numpy().view(np.uint64)[:] = _lib.ints_to_limbs([1, r, 0, 1, s, s, r * s % R, r])
        self.d_consts.copy_(self.h_consts, non_blocking=True)
        self.ext_a[m:] = self.d_consts[0:3]
        self.ext_b2[m:] = self.d_consts[3:5]
        if not self.bound:
            self.ext_b1[m:] = _dev(_lib.ints_to_limbs([0, 0, 1]))
        # u_A, u_B, u_C = coefficient forms (the reference's R.A etc.): 3 inverse NTTs, written where the MSMs read them (the
        # transforms run from one buffer to another: zk_ntt_dev_padded -- no copies)
        for src, dst in ((d_a, ua), (d_b, ub), (d_c, d_c)):
            self.ntt.run_padded(src.data_ptr(), dst.data_ptr(), m, True, None, st)
        self.ext_b2[:m].copy_(ub)
        # H = (A*B - C) / Z on the coset 5*H: 3 coset NTTs + pointwise quotient + 1 coset inverse NTT.  The transforms go
        # FIRST: an accumulate kernel fills every wavefront slot a CU frees, so NTT workgroups queued behind an MSM would
        # wait for its whole grid, the H query would start last and finish alone.
        for src, dst in ((ua, ca), (ub, cb), (d_c, cc)):
            self.ntt.run_padded(src.data_ptr(), dst.data_ptr(), m, False, COSET_SHIFT, st)
        fr_quotient(h.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), self.zinv, m, st)
        self.ntt.run(h.data_ptr(), True, COSET_SHIFT, st)
        if self.bound:
            # proving.py:47-75 with the +-r*s*delta terms cancelled:  proof_C = s*A + r*(beta*G1 + MSM(u_B, sigma1_2)) + L + H, and
            # s*A = s*alpha*G1 + MSM(s*u_A, sigma1_2) + r*s*delta*G1 is a combination of the same bases: the whole of proof_C is ONE
            # MSM with the scalars (s*u_A + r*u_B | s, r*s, r | w | h) -- no two-point combination (a pipeline of its own and a host
            # round trip) behind the last bucket reduction, and proof_C does not wait for proof_A.  Its scalars are put together
            # BEFORE any MSM is submitted: a small kernel queued behind an accumulate grid waits for that whole grid.
            sc = self.sc_c
            FrVec.lincomb(sc.data_ptr(), [ua.data_ptr(), ub.data_ptr()], [s, r], m, stream=st)   # s * u_A + r * u_B
            sc[m:m + 3] = self.d_consts[5:8]
            sc[self.off14:self.off15].copy_(d_w)                                         # placeholders at public wires are infinity
        # The MSMs, each in its own workspace and stream (the G1 plan keeps three in flight).  The G2 one leads: its long,
        # latency-bound bucket reduction then runs beside the G1 accumulate kernels instead of alone.
        t_b2 = self._msm(self.g2, self.ext_b2, crs.d_s22, 0, m + 2, st)                  # beta + B(x) + s*delta in G2
        t_a = self._msm(self.g1, self.ext_a, crs.d_s12, 0, m + 3, st)                    # alpha + A(x) + r*delta
        if self.bound:
            t_c = self.g1.submit_bound(sc.data_ptr(), 0, self.n_c, st)
            proof_b = self._pt(self.g2, self.g2.collect_limbs(t_b2))                     # proving.py:35-45
            proof_a = self._pt(self.g1, self.g1.collect_limbs(t_a))                      # proving.py:23-33
            proof_c = self._pt(self.g1, self.g1.collect_limbs(t_c))
            return proof_a, proof_b, proof_c, h
        t_b1 = self._msm(self.g1, self.ext_b1, crs.d_s12, 0, m + 3, st)                  # beta + B(x) in G1
        t_h = self._msm(self.g1, h, crs.d_s15, self.off15, m - 1, st)
        proof_a = self._pt(self.g1, self.g1.collect_limbs(t_a))                          # proving.py:23-33
        t_l = self._msm(self.g1, d_w, crs.d_s14, self.off14, W, st)                      # placeholders at public wires are infinity
        msm_b1 = self._pt(self.g1, self.g1.collect_limbs(t_b1))
        msm_h = self._pt(self.g1, self.g1.collect_limbs(t_h))
        msm_l = self._pt(self.g1, self.g1.collect_limbs(t_l))
        proof_b = self._pt(self.g2, self.g2.collect_limbs(t_b2))                         # proving.py:35-45
        # proving.py:47-75 with the +-r*s*delta terms cancelled:  s*A + r*(beta*G1 + MSM(u_B, sigma1_2)) + L + H
        proof_c = msm_g1([s, r, 1, 1], [proof_a, msm_b1, msm_l, msm_h])
        return proof_a, proof_b, proof_c, h

    def _msm(self, plan, scalars, points, first, count, st):
        """Submit one query MSM: over the bound table when the CRS is bound (first = offset of the query in it)."""
        if self.bound:
            return plan.submit_bound(scalars.data_ptr(), first, count, st)
        return plan.submit(scalars.data_ptr(), points.data_ptr(), count, st)

    def load_r1cs(self, csr):
        """Uploads the R1CS (dict name -> CSR triple, see ChainCircuit.r1cs_csr) for prove_from_witness."""
        import torch
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()
        self.r1cs = {k: tuple(up(x) for x in v) for k, v in csr.items()}
        self.abc = [torch.empty((self.m, 4), dtype=torch.int64, device="cuda") for _ in range(3)]

    def prove_from_witness(self, d_w, r, s, stream=None):
        """Witness (W, 4) on the device -> proof: the scalar collapse A.w, B.w, C.w (zk_fr_spmv_dev; the reference's
        proving.py:27-31 does it in the group), then prove()."""
        import torch
        from ..device import fr_spmv
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        for name, out in zip("ABC", self.abc):
            rp, col, vals = self.r1cs[name]
            fr_spmv(rp.data_ptr(), col.data_ptr(), vals.data_ptr(), d_w.data_ptr(), out.data_ptr(), self.m, st)
        return self.prove(self.abc[0], self.abc[1], self.abc[2], d_w, r, s, stream)

    @staticmethod
    def _pt(plan, res):
        limbs, inf = res
        if inf:
            return None
        from ..field import limbs_to_g2
        return (limbs_to_g1(limbs) if plan.group == _lib.GROUP_G1 else limbs_to_g2(limbs))[0]


class ShardedScaleProver(ScaleProver):
    """The same proof with every MSM sharded over the ranks of a process group by contiguous point chunks (one process
    per GPU): rank g multiplies its slice of each query by the matching slice of the scalars,
    the five XYZZ partials travel in ONE all-gather (96 limbs per rank) and are folded in rank order on every rank.  The
    transforms are cheap next to the MSMs and run replicated.  (A deployment would keep only its slices of the CRS in HBM;
    here every rank indexes into the full arrays.)"""

    def __init__(self, crs, group=None, device=None):
        import torch.distributed as dist
        super().__init__(crs)
        self.group, self.device = group, device
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def _slice(self, total):
        from ..distributed import shard_range
        return shard_range(total, self.rank, self.world)

    def prove(self, d_a, d_b, d_c, d_w, r, s, stream=None):
        import torch
        from ..distributed import all_gather_partials, fold_partials
        if stream is not None and stream != torch.cuda.current_stream().cuda_stream:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)):                # see ScaleProver.prove
                return self.prove(d_a, d_b, d_c, d_w, r, s, None)
        st = torch.cuda.current_stream().cuda_stream
        m, W, crs = self.m, self.W, self.crs
        r, s = r % R, s % R
        ca, cb, cc, h = self.scratch
        ua, ub = self.ext_a[:m], self.ext_b1[:m]
        ua.copy_(d_a)
        ub.copy_(d_b)
        self.ext_a[m:] = _dev(_lib.ints_to_limbs([1, r, 0]))
        self.ext_b1[m:] = _dev(_lib.ints_to_limbs([0, 0, 1]))
        self.ext_b2[m:] = _dev(_lib.ints_to_limbs([1, s]))
        for d in (ua, ub, d_c):
            self.ntt.run(d.data_ptr(), True, None, st)
        self.ext_b2[:m].copy_(ub)

        def submit(plan, scal, pts, total, point_bytes):
            lo, hi = self._slice(total)
            return plan.submit(scal.data_ptr() + 32 * lo, pts.data_ptr() + point_bytes * lo, hi - lo, st)

        ca.copy_(ua)                                                                  # transforms first, the G2 MSM leads: see ScaleProver.prove
        cb.copy_(ub)
        cc.copy_(d_c)
        for d in (ca, cb, cc):
            self.ntt.run(d.data_ptr(), False, COSET_SHIFT, st)
        fr_quotient(h.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), self.zinv, m, st)
        self.ntt.run(h.data_ptr(), True, COSET_SHIFT, st)
        t_b2 = submit(self.g2, self.ext_b2, crs.d_s22, m + 2, 128)
        t_a = submit(self.g1, self.ext_a, crs.d_s12, m + 3, 64)
        t_b1 = submit(self.g1, self.ext_b1, crs.d_s12, m + 3, 64)
        t_h = submit(self.g1, h, crs.d_s15, m - 1, 64)
        p_a = self.g1.collect_partial(t_a)
        t_l = submit(self.g1, d_w, crs.d_s14, W, 64)
        p_b1, p_h, p_l = self.g1.collect_partial(t_b1), self.g1.collect_partial(t_h), self.g1.collect_partial(t_l)
        mine = np.concatenate([p_a, p_b1, p_l, p_h, self.g2.collect_partial(t_b2)])    # 4 * 16 + 32 limbs
        everyone = all_gather_partials(mine, device=self.device, group=self.group)  # (world, 96)
        g1_parts = [fold_partials(_lib.GROUP_G1, np.ascontiguousarray(everyone[:, 16 * k:16 * (k + 1)])) for k in range(4)]
        proof_a, msm_b1, msm_l, msm_h = g1_parts
        proof_b = fold_partials(_lib.GROUP_G2, np.ascontiguousarray(everyone[:, 64:96]))
        proof_c = msm_g1([s, r, 1, 1], [proof_a, msm_b1, msm_l, msm_h])
        return proof_a, proof_b, proof_c, h


def closed_form_scalars(crs, witness, r, s):
    """(A, B, C) in F_r with proof_A = A*G1, proof_B = B*G2, proof_C = C*G1  (zkp/groth16/test.py:303-325)."""
    t = crs.toxic
    W = crs.circuit.num_wires
    a_x = sum(witness[i] * crs.Ax[i] for i in range(W)) % R
    b_x = sum(witness[i] * crs.Bx[i] for i in range(W)) % R
    c_x = sum(witness[i] * crs.Cx[i] for i in range(W)) % R
    A = (t["alpha"] + a_x + r * t["delta"]) % R
    B = (t["beta"] + b_x + s * t["delta"]) % R
    h_x = (a_x * b_x - c_x) % R * pow(crs.Zx, -1, R) % R
    priv = sum(witness[i] * crs.l_scalars[i] for i in range(W)) % R   # already divided by delta, public entries are 0
    C = (priv + h_x * crs.Zx % R * pow(t["delta"], -1, R) + A * s + B * r - r * s * t["delta"]) % R
    return A, B, C
