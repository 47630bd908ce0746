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
size is checked by the tests and by bench.py against closed-form scalars the ORACLE computes from the known toxic waste
(oracle/scale_ref.py; zkp/groth16/test.py:303-325: proof_A == A*G1, proof_B == B*G2, proof_C == C*G1).
"""
import numpy as np

from .. import _lib
from ..device import FrVec, MsmPlan, NttPlan, fr_quotient
from ..field import CURVE_ORDER as R, G1, G2, fixed_base_mul, g1_to_limbs, g2_to_limbs, msm_g1, limbs_to_g1

COSET_SHIFT = 5  # the reference's coset generator (zkp/plonk/utils.py:166-167)


from .circuits import BoolChainCircuit, ChainCircuit  # noqa: F401  (the synthetic R1CS generators live in circuits.py)


def _dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).cuda()


SPLIT = 1024   # longest run of entries one thread of the sparse mat-vec walks (see _transposed_times)


def _run_pointers(lens):
    """lens[i] consecutive entries belong to group i -> (ptr of the runs of at most SPLIT entries the groups are cut into, runs per group)."""
    import torch
    dev = lens.device
    starts = torch.cumsum(lens, 0) - lens
    runs = (lens + SPLIT - 1) // SPLIT
    run_first = torch.cumsum(runs, 0) - runs
    n_runs = int(runs.sum())
    group_of_run = torch.repeat_interleave(torch.arange(lens.shape[0], device=dev), runs)
    ptr = torch.empty(n_runs + 1, dtype=torch.int64, device=dev)
    ptr[:n_runs] = starts[group_of_run] + SPLIT * (torch.arange(n_runs, device=dev) - run_first[group_of_run])
    ptr[n_runs] = int(lens.sum())
    return ptr, runs


def _transposed_times(csr, num_wires, d_vec):
    """M^T v on the device for one R1CS matrix M (m x W, CSR on the host) and a device vector v of m elements: the per-wire sums
    sum_k M[k][i] v[k].  The transpose is a stable sort of the entries by column (torch index work; the values keep their limb
    form).  zk_fr_spmv_dev gives a row to ONE thread, and a wire such as `one` may sit in every constraint -- a row of 2^20 entries
    took 0.7 s -- so the entries of a wire are cut into runs of at most SPLIT (first product: one partial sum per run), and all-ones
    matrices then add the partial sums up, again at most SPLIT per thread, level by level until one sum per wire is left.
    Runs on torch's current stream (the temporaries of the loop are recycled by the caching allocator in that stream's order)."""
    import torch
    from ..device import fr_spmv
    stream = torch.cuda.current_stream().cuda_stream
    row_ptr, col, vals = csr
    m = row_ptr.shape[0] - 1
    dev = d_vec.device
    d_col = torch.from_numpy(col.astype(np.int64)).to(dev)
    counts = torch.from_numpy(np.diff(row_ptr.astype(np.int64))).to(dev)
    d_row = torch.repeat_interleave(torch.arange(m, device=dev), counts)
    order = torch.sort(d_col, stable=True).indices
    i32 = lambda t: t.to(torch.int32).contiguous()
    new = lambda rows: torch.empty((max(rows, 1), 4), dtype=torch.int64, device=dev)
    one_row = _dev(np.array([[1, 0, 0, 0]], dtype=np.uint64))
    lens = torch.bincount(d_col, minlength=num_wires)                      # entries per wire
    ptr, groups = _run_pointers(lens)
    n = ptr.shape[0] - 1
    cur = new(n)
    # (every operand is held in a name until the call has been issued: a temporary dropped while the argument list is still being
    # built hands its block back to the allocator, and the next temporary of the same list may be given it)
    p32, c32, v_t = i32(ptr), i32(d_row[order]), _dev(vals)[order].contiguous()
    fr_spmv(p32.data_ptr(), c32.data_ptr(), v_t.data_ptr(), d_vec.data_ptr(), cur.data_ptr(), n, stream)
    while True:                                                            # cur: n partial sums, groups[i] consecutive ones per wire
        last = int(groups.max()) <= SPLIT
        if last:
            ptr = torch.zeros(num_wires + 1, dtype=torch.int64, device=dev)
            ptr[1:] = torch.cumsum(groups, 0)
            rows = num_wires
        else:
            ptr, groups = _run_pointers(groups)
            rows = ptr.shape[0] - 1
        nxt = new(rows)
        p32, c32, v_t = i32(ptr), i32(torch.arange(n, device=dev)), one_row.repeat(max(n, 1), 1)
        fr_spmv(p32.data_ptr(), c32.data_ptr(), v_t.data_ptr(), cur.data_ptr(), nxt.data_ptr(), rows, stream)
        cur, n = nxt, rows
        if last:
            return cur


def crs_exponents(circuit, alpha, beta, delta, x):
    """The discrete logarithms of the CRS queries (zkp/groth16/setup.py:18-60 over the roots-of-unity QAP) as DEVICE vectors,
    produced by the F_r vector kernels and never visiting the host:
        w^k                       zk_fr_scale_powers_dev on a vector of ones
        1 / (x - w^k)             prefix and suffix product scans (zk_fr_scan_dev) and ONE host inversion of the total
        L_k(x)                    (x^m - 1) / m * w^k / (x - w^k)
        A_i(x), B_i(x), C_i(x)    the TRANSPOSED sparse R1CS matrices times L (zk_fr_spmv_dev, _transposed_times)
        lq[i]                     (beta A_i + alpha B_i + C_i)(x) / delta, zero at the public wires (setup.py:42-54)
        powers[j] = x^j,  hq[k] = x^k Z(x) / delta for k < m - 1 and 0 for k = m - 1 (setup.py:18-23, 56-60, 65-69)
    -> dict(powers (m, 4), lq (W, 4), hq (m, 4), qap {"A" | "B" | "C": (W, 4)}, Zx).
    Everything runs on torch's CURRENT stream: the torch temporaries in here are allocated and freed on it, so kernels launched on
    any other stream could still be reading a block the caching allocator has already handed out again."""
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    m, W = circuit.m, circuit.num_wires
    zx = (pow(x, m, R) - 1) % R
    if zx == 0 or delta == 0:
        raise ValueError("the toxic x must lie outside the evaluation domain and delta must be non-zero")
    dinv = pow(delta, -1, R)
    fv = FrVec()
    new = lambda rows: torch.empty((rows, 4), dtype=torch.int64, device="cuda")
    one_row = _dev(np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
    ones = one_row.repeat(m, 1)
    roots = ones.clone()
    fv.scale_powers(roots.data_ptr(), m, pow(5, (R - 1) // m, R), stream)             # w^k
    # 1 / (x - w^k) for all k: pre[k] = prod_{j<k} d_j, suf[k] = prod_{j>=k} d_j, inverse = pre[k] * suf[k+1] / total
    pre, suf = new(m + 1), new(m + 1)
    pre[0] = one_row
    suf[m] = one_row
    FrVec.lincomb(pre[1:].data_ptr(), [roots.data_ptr()], [R - 1], m, constant=x, stream=stream)      # d_k = x - w^k
    suf[:m].copy_(pre[1:])
    fv.scan(pre[1:].data_ptr(), m, product=True, reverse=False, stream=stream)
    fv.scan(suf.data_ptr(), m, product=True, reverse=True, stream=stream)
    total = _lib.limbs_to_ints(pre[m:m + 1].cpu().numpy().view(np.uint64))[0]
    lag = new(m)
    FrVec.mul(lag.data_ptr(), pre.data_ptr(), suf[1:].data_ptr(), m, stream)
    FrVec.mul(lag.data_ptr(), lag.data_ptr(), roots.data_ptr(), m, stream)
    FrVec.lincomb(lag.data_ptr(), [lag.data_ptr()], [zx * pow(m, -1, R) % R * pow(total, -1, R) % R], m, stream=stream)   # L_k(x)
    qap = {name: _transposed_times(csr, W, lag) for name, csr in circuit.r1cs_csr().items()}   # M_i(x) = sum_k M[k][i] L_k(x)
    lq = new(W)
    FrVec.lincomb(lq.data_ptr(), [qap[k].data_ptr() for k in "ABC"], [beta * dinv % R, alpha * dinv % R, dinv], W, stream=stream)
    lq[torch.from_numpy(np.array(circuit.pub, dtype=np.int64)).cuda()] = 0
    powers = ones
    fv.scale_powers(powers.data_ptr(), m, x, stream)
    hq = torch.zeros((m, 4), dtype=torch.int64, device="cuda")
    if m > 1:
        FrVec.lincomb(hq.data_ptr(), [powers.data_ptr()], [zx * dinv % R], m - 1, stream=stream)
    torch.cuda.synchronize()
    fv.close()
    return dict(powers=powers, lq=lq, hq=hq, qap=qap, Zx=zx)


class ScaleCRS:
    """Device-resident CRS in the reference's sigma layout (zkp/groth16/setup.py:15-69), built ON the device.

    The reference evaluates every wire polynomial at the toxic x with Python loops and multiplies G by each exponent in turn;
    at 2^20 constraints that shape costs seconds of interpreter time around kernels that take milliseconds.  Here the exponents
    come from crs_exponents (F_r vector kernels) and the points from the fixed-base batch kernels on device buffers
    (zk_fixed_base_g1_dev / _g2_dev): 0.1 s for 2^20 constraints where the Python loops took 3.5 s."""

    def __init__(self, circuit, alpha, beta, gamma, delta, x_val, keep_toxic=False):
        """keep_toxic: keep the toxic waste and the exponent vectors (the discrete logarithms of the queries) on the object -- for
        tests that check the key against closed forms; a key generator drops them (as DeviceSRS.generate does with tau)."""
        import torch
        self.circuit = circuit
        m, W = circuit.m, circuit.num_wires
        al, be, ga, de, x = (v % R for v in (alpha, beta, gamma, delta, x_val))
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        ex = crs_exponents(circuit, al, be, de, x)
        if keep_toxic:
            self.toxic = dict(alpha=al, beta=be, gamma=ga, delta=de, x=x)
            self.Zx, self.d_qap, self.d_l_scalars = ex["Zx"], ex["qap"], ex["lq"]
        # sigma1_1 / sigma2_1 (setup.py:15-16, 62-63)
        self.sigma1_1 = fixed_base_mul(G1, [al, be, de])
        self.sigma2_1 = fixed_base_mul(G2, [be, ga, de])
        g1, g2 = g1_to_limbs([G1]), g2_to_limbs([G2])
        # query arrays with the constant terms appended, so that a proof element is ONE device MSM:
        #   G1: sigma1_2 | alpha*G1 | delta*G1 | beta*G1        G2: sigma2_2 | beta*G2 | delta*G2
        self.d_s12 = torch.empty((m + 3, 8), dtype=torch.int64, device="cuda")
        self.d_s22 = torch.empty((m + 2, 16), dtype=torch.int64, device="cuda")
        self.d_s14 = torch.empty((W, 8), dtype=torch.int64, device="cuda")
        self.d_s15 = torch.empty((max(m - 1, 0), 8), dtype=torch.int64, device="cuda")
        _lib.check(lib.zk_fixed_base_g1_dev(_lib.ptr(g1), ex["powers"].data_ptr(), m, self.d_s12.data_ptr(), st))
        _lib.check(lib.zk_fixed_base_g2_dev(_lib.ptr(g2), ex["powers"].data_ptr(), m, self.d_s22.data_ptr(), st))
        _lib.check(lib.zk_fixed_base_g1_dev(_lib.ptr(g1), ex["lq"].data_ptr(), W, self.d_s14.data_ptr(), st))   # exponent 0 -> infinity (zeros)
        if m > 1:
            _lib.check(lib.zk_fixed_base_g1_dev(_lib.ptr(g1), ex["hq"].data_ptr(), m - 1, self.d_s15.data_ptr(), st))
        self.d_s12[m:] = _dev(g1_to_limbs([self.sigma1_1[0], self.sigma1_1[2], self.sigma1_1[1]]))
        self.d_s22[m:] = _dev(g2_to_limbs([self.sigma2_1[0], self.sigma2_1[2]]))
        torch.cuda.synchronize()


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
        self.a_chunked = self.m + 3 > (1 << (chunk_log or 22))
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
        self.profile = None     # set_profiling(True): a dict the next proof fills with the GPU time of its parts (HIP events)

    def set_profiling(self, enable):
        """With profiling on, a proof (bound CRS) records where its GPU time goes: the span of the transform / quotient section on the
        caller's stream (torch events) and, per MSM, the plan's stage spans {prepare, sort, accumulate, reduce}
        (zk_msm_plan_stage_ms) with the MSMs run one at a time, so that a span holds that MSM's kernels only.  The sum is the
        proof's kernel time; the pipelined wall clock of an unprofiled proof is measured separately."""
        self.g1.set_profiling(enable)
        self.g2.set_profiling(enable)
        self.profile = {} if enable else None

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
        # transforms run from one buffer to another -- no copies -- and the three of a group share one launch per pass:
        # zk_ntt_dev_multi; their workgroups share the chip, 0.31 ms instead of 0.36 per group at 2^20)
        if self.profile is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
        self.ntt.run_multi([(src.data_ptr(), dst.data_ptr()) for src, dst in ((d_a, ua), (d_b, ub), (d_c, d_c))], m, True, None, st)
        self.ext_b2[:m].copy_(ub)
        # H = (A*B - C) / Z on the coset 5*H: 3 coset NTTs + pointwise quotient + 1 coset inverse NTT.  The transforms go
        # FIRST: an accumulate kernel fills every wavefront slot a CU frees, so NTT workgroups queued behind an MSM would
        # wait for its whole grid, the H query would start last and finish alone.
        self.ntt.run_multi([(src.data_ptr(), dst.data_ptr()) for src, dst in ((ua, ca), (ub, cb), (d_c, cc))], m, False, COSET_SHIFT, st)
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
        if self.profile is not None:
            ev[1].record()
        if self.profile is not None and self.bound:
            # profiling: the three MSMs ONE AT A TIME (submit, collect, next), so that every stage span is the time of that MSM's own
            # kernels and the sum is the proof's kernel time; the proof is the same, its wall clock is not the pipelined one
            proof_b = self._pt(self.g2, self._collect(self.g2, self._msm(self.g2, self.ext_b2, crs.d_s22, 0, m + 2, st), "msm_B_g2"))
            proof_a = self._pt(self.g1, self._collect(self.g1, self._msm(self.g1, self.ext_a, crs.d_s12, 0, m + 3, st), "msm_A_g1"))
            proof_c = self._pt(self.g1, self._collect(self.g1, self.g1.submit_bound(self.sc_c.data_ptr(), 0, self.n_c, st), "msm_C_g1_merged"))
            torch.cuda.synchronize()
            self.profile["transforms_and_quotient_ms"] = round(ev[0].elapsed_time(ev[1]), 4)
            return proof_a, proof_b, proof_c, h
        t_b2 = self._msm(self.g2, self.ext_b2, crs.d_s22, 0, m + 2, st)                  # beta + B(x) + s*delta in G2
        t_a = self._msm(self.g1, self.ext_a, crs.d_s12, 0, m + 3, st)                    # alpha + A(x) + r*delta
        if self.bound:
            # from 2^22 constraints on the A query (m + 3 bases) is itself more than one chunk, and a plan takes ONE chunked submission
            # at a time: it is collected before the merged query goes in
            proof_a = self._pt(self.g1, self.g1.collect_limbs(t_a)) if self.a_chunked else None
            t_c = self.g1.submit_bound(sc.data_ptr(), 0, self.n_c, st)
            proof_b = self._pt(self.g2, self.g2.collect_limbs(t_b2))                     # proving.py:35-45
            if not self.a_chunked:
                proof_a = self._pt(self.g1, self.g1.collect_limbs(t_a))                  # proving.py:23-33
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

    def _collect(self, plan, ticket, name):
        res = plan.collect_limbs(ticket)
        if self.profile is not None:
            self.profile[name] = dict(zip(("prepare", "sort", "accumulate", "reduce"), (round(v, 4) for v in plan.stage_ms())))
        return res

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
        # A, B, C stacked into ONE CSR matrix of 3m rows: the three mat-vecs are one launch whose workgroups share the chip (three
        # launches of 4096 latency-bound workgroups each ran one after the other at the start of every proof)
        rp, col, vals, base = [np.zeros(1, dtype=np.uint32)], [], [], 0
        for k in "ABC":
            r_k, c_k, v_k = (np.ascontiguousarray(x) for x in csr[k])
            rp.append(r_k[1:].astype(np.uint32) + np.uint32(base))
            col.append(c_k)
            vals.append(v_k)
            base += int(r_k[-1])
        self.r1cs_stacked = (up(np.concatenate(rp)), up(np.concatenate(col)), up(np.concatenate(vals)))
        self.abc_all = torch.empty((3 * self.m, 4), dtype=torch.int64, device="cuda")
        self.abc = [self.abc_all[i * self.m:(i + 1) * self.m] for i in range(3)]

    def prove_from_witness(self, d_w, r, s, stream=None):
        """Witness (W, 4) on the device -> proof: the scalar collapse A.w, B.w, C.w (zk_fr_spmv_dev; the reference's
        proving.py:27-31 does it in the group), then prove()."""
        import torch
        from ..device import fr_spmv
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        timed = self.profile is not None and stream is None
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rp, col, vals = self.r1cs_stacked
        fr_spmv(rp.data_ptr(), col.data_ptr(), vals.data_ptr(), d_w.data_ptr(), self.abc_all.data_ptr(), 3 * self.m, st)
        if timed:
            e1.record()
        res = self.prove(self.abc[0], self.abc[1], self.abc[2], d_w, r, s, stream)
        if timed:
            self.profile["sparse_matvecs_ms"] = round(e0.elapsed_time(e1), 4)
        return res

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
        self.ntt.run_multi([(d.data_ptr(), d.data_ptr()) for d in (ua, ub, d_c)], m, True, None, st)   # one launch per pass for the three
        self.ext_b2[:m].copy_(ub)

        def submit(plan, scal, pts, total, point_bytes):
            lo, hi = self._slice(total)
            return plan.submit(scal.data_ptr() + 32 * lo, pts.data_ptr() + point_bytes * lo, hi - lo, st)

        ca.copy_(ua)                                                                  # transforms first, the G2 MSM leads: see ScaleProver.prove
        cb.copy_(ub)
        cc.copy_(d_c)
        self.ntt.run_multi([(d.data_ptr(), d.data_ptr()) for d in (ca, cb, cc)], m, False, COSET_SHIFT, st)
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
