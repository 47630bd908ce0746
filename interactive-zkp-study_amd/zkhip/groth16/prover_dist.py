"""Groth16 at scale with NOTHING replicated across the GPUs of a node (SURVEY.md section 8 rows E1 + E2 together).

`ShardedScaleProver` (prover_ntt.py) shards the MSMs by contiguous chunks and runs the seven transforms on every rank: at 8 ranks
the replicated transforms (~1.2 ms at 2^20 constraints) are an Amdahl floor under ~1.1 ms of sharded MSM work, and every rank holds
the whole CRS.  Here every vector of the proof lives in the block-cyclic layout of the four-step transform (zkhip.distributed.DistNtt):

    evaluations   (A.w, B.w, C.w per constraint, the quotient on the coset)   BC(n1): rank r owns the constraints K with
                                                                              (K mod n1) in its 1/R slice, stored [k1 local][k2]
    coefficients  (u_A, u_B, u_C, H)                                          BC(n2): rank r owns the coefficients i with
                                                                              (i mod n2) in its slice, stored [j2 local][j1]
so that
  * the sparse mat-vecs run over the rank's OWN constraint rows (zk_fr_spmv_dev on a row subset of the R1CS),
  * each of the seven transforms is ONE all-to-all of m / R^2 elements per pair of ranks plus 1/R of the butterflies
    (DistNtt.forward / inverse; the coset shift is a pointwise product with a precomputed BC-ordered power vector),
  * the pointwise quotient is layout-agnostic,
  * and the MSMs run over the rank's coefficients against ITS slice of the queries -- sigma1_2, sigma2_2, sigma1_5 generated at
    setup for exactly the coefficient indices the rank owns (1/R of the fixed-base work and of the HBM), sigma1_4 for a contiguous
    slice of the wires -- down to three XYZZ partial sums per rank, exchanged in ONE all-gather of 64 limbs and folded in rank order
    (zk_msm_fold_partials), as in zkhip.distributed.sharded_msm.
Nothing but the witness (an input) is held twice.  The proof is bit-identical to the single-GPU prover's and checked against the
oracle's closed form (tests/test_gpu_groth16_dist.py: 1 rank, and 2 and 4 ranks rehearsed on one GPU over gloo).

Reference: the same formulas as prover_ntt.ScaleProver (zkp/groth16/proving.py:23-75, setup.py:15-69)."""
import numpy as np

from .. import _lib
from ..device import FrVec, MsmPlan, fr_quotient, fr_spmv
from ..distributed import DistNtt, all_gather_partials, fold_partials, shard_range
from ..field import CURVE_ORDER as R, G1, G2, fixed_base_mul, g1_to_limbs, g2_to_limbs
from .prover_ntt import COSET_SHIFT, _dev, crs_exponents


def _rows_subset(csr, rows):
    """CSR of the given rows (in that order) of a host CSR matrix; numpy only."""
    row_ptr, col, vals = csr
    rp = row_ptr.astype(np.int64)
    lens = (rp[1:] - rp[:-1])[rows]
    new_ptr = np.zeros(rows.shape[0] + 1, dtype=np.int64)
    np.cumsum(lens, out=new_ptr[1:])
    take = np.repeat(rp[rows] - new_ptr[:-1], lens) + np.arange(int(new_ptr[-1]))
    return new_ptr.astype(np.uint32), col[take], vals[take]


class DistScaleCRS:
    """This rank's share of the CRS (see the module header).  Every rank computes the exponent vectors (cheap F_r vector work);
    the fixed-base batches -- the expensive part of key generation -- and the stored points cover the rank's slices only."""

    def __init__(self, circuit, alpha, beta, gamma, delta, x_val, group=None, keep_toxic=False):
        """keep_toxic: keep the toxic waste on the object (tests only; see ScaleCRS)."""
        import torch
        import torch.distributed as dist
        self.circuit, self.group = circuit, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        m, W, Rk, rank = circuit.m, circuit.num_wires, self.world, self.rank
        al, be, ga, de, x = (v % R for v in (alpha, beta, gamma, delta, x_val))
        if keep_toxic:
            self.toxic = dict(alpha=al, beta=be, gamma=ga, delta=de, x=x)
        self.dn = DistNtt(circuit.log_m, group=group)
        n1, n2, c, k = self.dn.n1, self.dn.n2, self.dn.c, self.dn.k
        self.cn = c * n1                                                   # coefficients (and evaluations: k * n2, the same number) per rank
        st = torch.cuda.current_stream().cuda_stream
        ex = crs_exponents(circuit, al, be, de, x)
        bc2 = lambda full: full.view(n1, n2, 4)[:, rank * c:(rank + 1) * c].transpose(0, 1).contiguous().view(self.cn, 4)   # [j2 local][j1]
        lib = _lib.load()
        g1, g2 = g1_to_limbs([G1]), g2_to_limbs([G2])
        pw, hq = bc2(ex["powers"]), bc2(ex["hq"])
        self.w_lo, self.w_hi = shard_range(W, rank, Rk)                    # the rank's wires of the L query
        nw = self.w_hi - self.w_lo
        # G1 bases of this rank: sigma1_2 slice | alpha, delta, beta (rank 0; infinity elsewhere) | sigma1_4 slice | sigma1_5 slice
        self.off_const, self.off14, self.off15 = self.cn, self.cn + 3, self.cn + 3 + nw
        self.n_g1 = self.off15 + self.cn
        self.d_g1 = torch.zeros((self.n_g1, 8), dtype=torch.int64, device="cuda")
        _lib.check(lib.zk_fixed_base_g1_dev(_lib.ptr(g1), pw.data_ptr(), self.cn, self.d_g1.data_ptr(), st))
        if nw:
            lq_mine = ex["lq"][self.w_lo:self.w_hi].contiguous()
            _lib.check(lib.zk_fixed_base_g1_dev(_lib.ptr(g1), lq_mine.data_ptr(), nw, self.d_g1[self.off14:].data_ptr(), st))
        _lib.check(lib.zk_fixed_base_g1_dev(_lib.ptr(g1), hq.data_ptr(), self.cn, self.d_g1[self.off15:].data_ptr(), st))
        # G2 bases: sigma2_2 slice | beta, delta (rank 0)
        self.d_g2 = torch.zeros((self.cn + 2, 16), dtype=torch.int64, device="cuda")
        _lib.check(lib.zk_fixed_base_g2_dev(_lib.ptr(g2), pw.data_ptr(), self.cn, self.d_g2.data_ptr(), st))
        if rank == 0:
            s11 = fixed_base_mul(G1, [al, be, de])                         # sigma1_1 (setup.py:15-16)
            s21 = fixed_base_mul(G2, [be, ga, de])                         # sigma2_1 (setup.py:62-63)
            self.d_g1[self.off_const:self.off14] = _dev(g1_to_limbs([s11[0], s11[2], s11[1]]))
            self.d_g2[self.cn:] = _dev(g2_to_limbs([s21[0], s21[2]]))
        # the coset shift 5^(+-i) of the rank's coefficients, in their storage order
        fv = FrVec()
        one_row = _dev(np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
        cos = []
        for base in (COSET_SHIFT, pow(COSET_SHIFT, -1, R)):
            full = one_row.repeat(m, 1)
            fv.scale_powers(full.data_ptr(), m, base, st)
            cos.append(bc2(full))
        torch.cuda.synchronize()
        fv.close()
        self.cos_fwd, self.cos_inv = cos
        # the rank's constraint rows, in BC(n1) storage order [k1 local][k2]: K = k1 + n1 * k2
        k1 = np.arange(rank * k, (rank + 1) * k, dtype=np.int64)
        rows = (k1[:, None] + n1 * np.arange(n2, dtype=np.int64)[None, :]).reshape(-1)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()
        self.r1cs = {name: tuple(up(v) for v in _rows_subset(csr, rows)) for name, csr in circuit.r1cs_csr().items()}


class DistScaleProver:
    """Proves against a DistScaleCRS; every rank calls prove() with the full witness on its device and gets the same proof."""

    def __init__(self, crs, device=None):
        import torch
        self.crs, self.device = crs, device
        self.cn, self.bound = crs.cn, crs.n_g1 > (1 << 17)
        self.g1 = MsmPlan(_lib.GROUP_G1, max(crs.n_g1, (1 << 17) + 1 if self.bound else 1))
        self.g2 = MsmPlan(_lib.GROUP_G2, max(crs.cn + 2, (1 << 17) + 1 if crs.cn + 2 > (1 << 17) else 1))
        self.bound2 = crs.cn + 2 > (1 << 17)
        st = torch.cuda.current_stream().cuda_stream
        if self.bound:
            self.g1.bind(crs.d_g1.data_ptr(), crs.n_g1, st)
        if self.bound2:
            self.g2.bind(crs.d_g2.data_ptr(), crs.cn + 2, st)
        new = lambda rows: torch.empty((rows, 4), dtype=torch.int64, device="cuda")
        self.ev = [new(self.cn) for _ in range(3)]                        # A.w, B.w, C.w on the rank's rows
        self.coset = [new(self.cn) for _ in range(3)]
        self.sc_a, self.sc_b2, self.sc_c = new(self.cn + 3), new(self.cn + 2), new(crs.n_g1)
        self.zinv = pow((pow(COSET_SHIFT, crs.circuit.m, R) - 1) % R, -1, R)   # 1 / Z_H on the coset 5 * H

    def _partial(self, plan, bound, scalars, points, first, count, st):
        if bound:
            return plan.collect_partial(plan.submit_bound(scalars.data_ptr(), first, count, st))
        return plan.run_partial(scalars.data_ptr(), points.data_ptr(), count, st)

    def prove(self, d_w, r, s):
        """d_w: (W, 4) witness on this rank's device -> (proof_A, proof_B, proof_C); identical on every rank."""
        import torch
        crs, cn, dn = self.crs, self.cn, self.crs.dn
        st = torch.cuda.current_stream().cuda_stream
        r, s = r % R, s % R
        shape_ev, shape_co = (dn.k, dn.n2, 4), (dn.c, dn.n1, 4)
        # per-constraint values on the rank's rows, then coefficient form: 3 inverse transforms (one all-to-all each)
        # (the three exchanges of a group are in flight together: B's local transform runs under A's all-to-all, C's under B's --
        # DistNtt.inverse_many / forward_many)
        for name, ev in zip("ABC", self.ev):
            rp, col, vals = crs.r1cs[name]
            fr_spmv(rp.data_ptr(), col.data_ptr(), vals.data_ptr(), d_w.data_ptr(), ev.data_ptr(), cn, st)
        coef = [u.reshape(cn, 4) for u in dn.inverse_many([ev.view(shape_ev) for ev in self.ev])]   # u_A, u_B, u_C: BC(n2), [j2 local][j1]
        ua, ub, uc = coef
        # H = (A B - C) / Z on the coset: shift, 3 forward transforms, pointwise quotient, inverse transform, shift back
        for u, buf in zip(coef, self.coset):
            FrVec.mul(buf.data_ptr(), u.data_ptr(), crs.cos_fwd.data_ptr(), cn, st)
        ca, cb, cc = (v.reshape(cn, 4) for v in dn.forward_many([buf.view(shape_co) for buf in self.coset]))
        fr_quotient(ca.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), self.zinv, cn, st)
        h = dn.inverse(ca.view(shape_ev)).reshape(cn, 4)
        FrVec.mul(h.data_ptr(), h.data_ptr(), crs.cos_inv.data_ptr(), cn, st)
        # scalars of the three MSMs behind this rank's bases (constant terms: alpha, delta, beta resp. beta, delta -- rank 0's points)
        consts = _dev(_lib.ints_to_limbs([1, r, 0, 1, s, s, r * s % R, r]))
        self.sc_a[:cn].copy_(ua)
        self.sc_a[cn:] = consts[0:3]
        self.sc_b2[:cn].copy_(ub)
        self.sc_b2[cn:] = consts[3:5]
        sc = self.sc_c                                                     # proof_C as ONE MSM, as in ScaleProver.prove
        FrVec.lincomb(sc.data_ptr(), [ua.data_ptr(), ub.data_ptr()], [s, r], cn, stream=st)
        sc[crs.off_const:crs.off14] = consts[5:8]
        sc[crs.off14:crs.off15].copy_(d_w[crs.w_lo:crs.w_hi])
        sc[crs.off15:].copy_(h)
        p_b2 = self._partial(self.g2, self.bound2, self.sc_b2, crs.d_g2, 0, cn + 2, st)
        p_a = self._partial(self.g1, self.bound, self.sc_a, crs.d_g1, 0, cn + 3, st)
        p_c = self._partial(self.g1, self.bound, sc, crs.d_g1, 0, crs.n_g1, st)
        mine = np.concatenate([p_a, p_c, p_b2])                            # 16 + 16 + 32 limbs
        if crs.world > 1:
            everyone = all_gather_partials(mine, device=self.device, group=crs.group)
        else:
            everyone = mine.reshape(1, -1)
        proof_a = fold_partials(_lib.GROUP_G1, np.ascontiguousarray(everyone[:, 0:16]))
        proof_c = fold_partials(_lib.GROUP_G1, np.ascontiguousarray(everyone[:, 16:32]))
        proof_b = fold_partials(_lib.GROUP_G2, np.ascontiguousarray(everyone[:, 32:64]))
        return proof_a, proof_b, proof_c
