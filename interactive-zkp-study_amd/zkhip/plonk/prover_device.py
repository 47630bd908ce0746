"""PLONK preprocessing and prover with every vector resident in HBM (SURVEY.md section 8 row f3 at scale).

Same protocol, transcript, blinding degrees and proof fields as zkhip/plonk/prover.py (which mirrors
zkp/plonk/prover/round1..5.py on Python lists and is what the reference-sized circuits use); here the columns,
selector / permutation polynomials and the SRS are device buffers and the rounds are sequences of backend calls:

  interpolation, coset evaluation      zk_ntt_dev_padded (NttPlan.run_padded: coefficients -> a larger domain, no zero fill / copy)
  commitments                          zk_msm_submit / collect (MsmPlan, up to three in flight)
  a + beta*id + gamma, r(x), ...       zk_fr_lincomb_dev, zk_fr_mul_dev
  round-3 quotient on the coset        zk_plonk_quotient_dev (one fused pass over 15 vectors)
  grand product z                      prefix products of the numerators, suffix products of the denominators (zk_fr_scan_dev)
  p(zeta) of all openings of a round   zk_fr_eval_dev (one pass over the coefficients)
  (p(x) - p(z)) / (x - z)              q[i] = z^-(i+1) * sum_{j>i} c[j] z^j: scale, suffix sums, scale

The host only hashes the transcript and handles the ~20 scalars between rounds.  With the same blinding scalars the
proof equals the list prover's bit for bit (tests/test_gpu_plonk_device.py); the reference's quirks (PI(x) = 0, no
blinding of t) are kept."""
import os
import secrets

import numpy as np

from .. import _lib
from ..device import FrVec, MsmPlan, NttPlan, plonk_perm_factors, plonk_quotient
from ..field import FR, CURVE_ORDER as R, get_root_of_unity, limbs_to_g1
from .permutation import K1, K2
from .prover import COSET_K, Proof, linearisation_scalars
from .prover.common import NOT_DIVISIBLE
from .transcript import Transcript

PAD = 8  # slack coefficients behind the n of every polynomial buffer (blinding adds up to 3, t_hi up to 6)


def _torch():
    import torch
    return torch


def _dev(limbs):
    return _torch().from_numpy(np.ascontiguousarray(limbs).view(np.int64)).cuda()


def _limbs(ints):
    return _lib.ints_to_limbs([int(v) % R for v in ints])


class DevicePlonk:
    """Circuit of n = 2^k gates given by its selector and permutation EVALUATIONS on the domain (limb arrays
    (n, 4) uint64, or device tensors (n, 4) int64: q_l, q_r, q_o, q_m, q_c, s_sigma1..3) and an SRS as a limb array / device
    tensor of G1 points (>= n + 6 rows)."""

    def __init__(self, selectors, sigmas, srs_g1_limbs):
        torch = _torch()
        n = selectors[0].shape[0]
        if n < 4 or n & (n - 1):
            raise ValueError("DevicePlonk: n must be a power of two >= 4")
        if srs_g1_limbs.shape[0] < n + 6:
            raise ValueError("DevicePlonk: the SRS must hold at least n + 6 powers")
        self.n, self.log_n = n, n.bit_length() - 1
        self.omega = get_root_of_unity(n)
        self.size = 1
        while self.size < 3 * n + 6:
            self.size <<= 1
        self.step = self.size // n
        self.st = torch.cuda.current_stream().cuda_stream
        self.fv = FrVec()
        self.ntt_n = NttPlan(self.log_n)
        self.ntt_big = NttPlan(self.size.bit_length() - 1)
        self.msm = MsmPlan(_lib.GROUP_G1, n + PAD)
        if torch.is_tensor(srs_g1_limbs):
            self.srs = torch.zeros((n + PAD, 8), dtype=torch.int64, device="cuda")
            rows = min(n + PAD, srs_g1_limbs.shape[0])
            self.srs[:rows] = srs_g1_limbs[:rows]
        else:
            self.srs = _dev(srs_g1_limbs[:n + PAD] if srs_g1_limbs.shape[0] >= n + PAD else np.concatenate(
                [srs_g1_limbs, np.zeros((n + PAD - srs_g1_limbs.shape[0], 8), dtype=np.uint64)]))
        # the SRS never changes: bind it (table of 2^(20 w) * [tau^i], 13 n bucket additions per commitment instead of 16 n)
        self.bound = n + PAD > (1 << 17)
        if self.bound:
            self.msm.bind(self.srs.data_ptr(), n + PAD, self.st)
        names = ("q_l", "q_r", "q_o", "q_m", "q_c", "s_sigma1", "s_sigma2", "s_sigma3")
        self.evals = {k: (v if torch.is_tensor(v) else _dev(v)) for k, v in zip(names, list(selectors) + list(sigmas))}
        # coefficient forms (8 inverse NTTs) and commitments (8 MSMs)
        self.coef, self.comm = {}, {}
        for k in names:
            self.coef[k] = self._interpolate(self.evals[k])
        for k in names:
            self.comm[k] = self._commit(self.coef[k], n)
        # identity labels omega^i (times 1, K1, K2 by coefficient) and everything the quotient needs on the coset k*H'
        one = np.zeros((n + PAD, 4), dtype=np.uint64)
        one[:, 0] = 1
        self.ones = _dev(one)
        ones = self.ones[:n].clone()
        self.fv.scale_powers(ones.data_ptr(), n, int(self.omega), self.st)
        self.ident = ones
        self.coset = {k: self._coset(self.coef[k]) for k in names}
        x_coef = self._zeros(n + PAD)
        x_coef[1:2] = _dev(_limbs([1]))
        self.coset["x"] = self._coset(x_coef)
        l1_coef = self._zeros(n + PAD)
        l1_coef[:n] = _dev(_limbs([pow(n, -1, R)]))              # L_1(x) = (x^n - 1) / (n (x - 1)) = (1/n) sum_j x^j  (one row, broadcast)
        self.coset["l1"] = self._coset(l1_coef)
        w_big = int(get_root_of_unity(self.size))
        zh_inv = [pow((pow(COSET_K * pow(w_big, i, R) % R, n, R) - 1) % R, -1, R) for i in range(self.step)]
        self.zh_inv = zh_inv                                     # 1 / Z_H: x^n has period `step` on the coset
        self.work = [self._zeros(self.size) for _ in range(6)]   # per-proof coset buffers: a, b, c, z, z(omega x), t
        # per-proof coefficient / scratch vectors, allocated once: prove() itself allocates nothing on the device, so the
        # caching allocator never has to find (or release and re-acquire) twenty 32 MB blocks in the middle of a proof
        self.buf = {k: self._zeros(n + PAD) for k in ("w0", "w1", "w2", "z", "num", "den", "tmp", "z_ev", "t0", "t1", "t2",
                                                      "r_poly", "numer", "dtmp", "q0", "q1")}
        self.small = self._zeros(16)

    # ---- helpers -------------------------------------------------------------------------------------------
    def _zeros(self, rows):
        return _torch().zeros((rows, 4), dtype=_torch().int64, device="cuda")

    def _interpolate(self, evals, out=None):
        """(n, 4) evaluations on the domain -> zero-padded coefficient buffer (n + PAD, 4) (`out` when given)."""
        if out is None:
            out = self._zeros(self.n + PAD)
        else:
            out[self.n:].zero_()
        self.ntt_n.run_padded(evals.data_ptr(), out.data_ptr(), self.n, True, None, self.st)   # evals -> out[:n], no copy
        return out

    def _interpolate_many(self, evals, outs):
        """_interpolate of up to four columns in one launch per pass (zk_ntt_dev_multi): the transforms share the chip."""
        for out in outs:
            out[self.n:].zero_()
        self.ntt_n.run_multi([(e.data_ptr(), o.data_ptr()) for e, o in zip(evals, outs)], self.n, True, None, self.st)
        return outs

    def _coset_many(self, coefs, outs):
        """_coset of up to four coefficient buffers of one length in one launch per pass."""
        m = min(coefs[0].shape[0], self.size)
        assert all(min(c.shape[0], self.size) == m for c in coefs)
        self.ntt_big.run_multi([(c.data_ptr(), o.data_ptr()) for c, o in zip(coefs, outs)], m, False, COSET_K, self.st)
        return outs

    def _coset(self, coef, out=None):
        """Coefficient buffer -> evaluations on the coset k*H' (size, 4), into `out` when given."""
        if out is None:
            out = _torch().empty((self.size, 4), dtype=_torch().int64, device="cuda")
        m = min(coef.shape[0], self.size)
        # the coefficients beyond m count as zero and are not read: neither a zero fill of the 4n-point buffer nor a copy into it
        self.ntt_big.run_padded(coef.data_ptr(), out.data_ptr(), m, False, COSET_K, self.st)
        return out

    def _commit(self, coef, count):
        limbs, inf = self.msm.run_limbs(coef.data_ptr(), None if self.bound else self.srs.data_ptr(), count, self.st)
        return None if inf else limbs_to_g1(limbs)[0]

    def _submit(self, coef, count):
        return self.msm.submit(coef.data_ptr(), None if self.bound else self.srs.data_ptr(), count, self.st)

    def _commit_many(self, items):
        """[(coef, count)] -> points; the MSMs are kept in flight together."""
        depth, pend, out = self.msm.max_in_flight(), [], []
        for coef, count in items:
            if len(pend) == depth:
                out.append(self.msm.collect_limbs(pend.pop(0)))
            pend.append(self._submit(coef, count))
        out += [self.msm.collect_limbs(t) for t in pend]
        return [None if inf else limbs_to_g1(limbs)[0] for limbs, inf in out]

    def _lin(self, out, ins, coeffs, n, constant=None):
        FrVec.lincomb(out.data_ptr(), [t.data_ptr() for t in ins], coeffs, n, constant, self.st)

    def _mul(self, out, a, b, n):
        FrVec.mul(out.data_ptr(), a.data_ptr(), b.data_ptr(), n, self.st)

    def _evaluate(self, coef, count, point):
        """p(point) = sum_i c_i point^i (polynomial.py:85-106)."""
        return self._evaluate_many([(coef, count)], point)[0]

    def _evaluate_many(self, items, point):
        """[(coef, count)] -> [p(point)]: one pass over the coefficients of all of them (zk_fr_eval_dev), all values back in one copy."""
        out = []
        for lo in range(0, len(items), 8):
            part = items[lo:lo + 8]
            res = self.small[:len(part)]
            self.fv.eval([(coef.data_ptr(), count) for coef, count in part], int(point) % R, res.data_ptr(), self.st)
            out += [FR(v) for v in _lib.limbs_to_ints(res.cpu().numpy().view(np.uint64))]
        return out

    def _divide_linear(self, coef, count, point, q):
        """Quotient of p(x) / (x - point) into q: count - 1 coefficients (the remainder p(point) is dropped; rows of q
        behind them are not defined)."""
        tmp = self.buf["dtmp"]
        tmp[:count] = coef[:count]
        z = int(point) % R
        self.fv.scale_powers(tmp.data_ptr(), count, z, self.st)
        self.fv.scan(tmp.data_ptr(), count, False, True, self.st)              # suffix sums of c_j z^j
        q[:count - 1] = tmp[1:count]
        zi = pow(z, -1, R)
        self.fv.scale_powers(q.data_ptr(), count - 1, zi, self.st)
        self._lin(q, [q], [zi], count - 1)
        return q

    def _blinded(self, coef, bl):
        """coef + blind(x) * (x^n - 1) in place (round1.py:92-108); bl: (k, 4) device tensor of the blinding scalars (uploaded once per
        proof: a host-to-device copy of pageable memory in the middle of a round holds the host until the stream has drained)."""
        n, k = self.n, bl.shape[0]
        self._lin(coef[:k], [coef[:k], bl], [1, R - 1], k)
        self._lin(coef[n:n + k], [coef[n:n + k], bl], [1, 1], k)
        return coef

    # ---- grand product -------------------------------------------------------------------------------------------
    def _accumulator(self, cols, be, ga):
        """z on the domain (permutation.py:89-137): z_0 = 1, z_{i+1} = z_i * num_i / den_i, as prefix products of the numerators
        times suffix products of the denominators over their total -- ONE inversion.  The reference divides row by row with
        inv0(0) = 0, so a zero denominator zeroes z from that row on and leaves the rows before it; a zero TOTAL (probability
        about n / r for Fiat-Shamir challenges; it also covers row n - 1, which the reference never divides by) cannot be
        inverted here, and that case takes the row-exact host path (_accumulator_exact), so z equals the reference's always."""
        n, fv, st = self.n, self.fv, self.st
        num, den = self._accumulator_factors(cols, be, ga)
        fv.scan(num.data_ptr(), n, True, False, st)                              # prod_{j<=i} num_j
        fv.scan(den.data_ptr(), n, True, True, st)                               # prod_{j>=i} den_j
        den_total = _lib.limbs_to_ints(den[:1].cpu().numpy().view(np.uint64))[0]
        if den_total == 0:
            return self._accumulator_exact(*self._accumulator_factors(cols, be, ga))   # the scans ran in place: rebuild the factors
        z_ev = self.buf["z_ev"][:n]
        z_ev[:1] = self.ones[:1]
        self._mul(z_ev[1:], num[:n - 1], den[1:], n - 1)                         # z_i = prod_{j<i} num_j / den_j
        self._lin(z_ev[1:], [z_ev[1:]], [pow(den_total, -1, R)], n - 1)
        return z_ev

    def _accumulator_factors(self, cols, be, ga):
        """Per row: num_i = prod_w (w_i + beta * k_w * omega^i + gamma), den_i = prod_w (w_i + beta * sigma_w(i) + gamma) -- one fused
        pass over the seven vectors (zk_plonk_perm_factors_dev; ten lincomb / product launches until round 5)."""
        n, B = self.n, self.buf
        num, den = B["num"][:n], B["den"][:n]
        sig = [self.evals["s_sigma%d" % k] for k in (1, 2, 3)]
        if os.environ.get("ZK_PLONK_UNFUSED_FACTORS"):                          # A/B runs and the test that compares the two forms
            return self._accumulator_factors_unfused(cols, be, ga)
        plonk_perm_factors(num.data_ptr(), den.data_ptr(), [t.data_ptr() for t in (cols[0], cols[1], cols[2], sig[0], sig[1], sig[2], self.ident)],
                           be, ga, n, self.st)
        return num, den

    def _accumulator_factors_unfused(self, cols, be, ga):
        """The same factors by linear combinations and products of whole vectors (the form used until round 5; kept as the
        cross-check of the fused kernel: tests/test_gpu_plonk_device.py)."""
        n, B = self.n, self.buf
        num, den, tmp = B["num"][:n], B["den"][:n], B["tmp"][:n]
        sig = [self.evals["s_sigma%d" % k] for k in (1, 2, 3)]
        for j, (col, idc) in enumerate(zip(cols, (1, int(K1), int(K2)))):
            self._lin(tmp, [col, self.ident], [1, be * idc % R], n, ga)          # w + beta * k * omega^i + gamma
            if j == 0:
                num.copy_(tmp)
            else:
                self._mul(num, num, tmp, n)
            self._lin(tmp, [col, sig[j]], [1, be], n, ga)                        # w + beta * sigma(i) + gamma
            if j == 0:
                den.copy_(tmp)
            else:
                self._mul(den, den, tmp, n)
        return num, den

    def _accumulator_exact(self, num, den):
        """The reference's loop on the host, from the per-row numerators / denominators: z_{i+1} = z_i * num_i * inv0(den_i)
        with inv0(0) = 0 (py_ecc's FQ division); one batch inversion over the non-zero denominators."""
        n = self.n
        nums = _lib.limbs_to_ints(num.cpu().numpy().view(np.uint64))
        dens = _lib.limbs_to_ints(den.cpu().numpy().view(np.uint64))
        pref = [1]
        for d in dens[:n - 1]:
            pref.append(pref[-1] * (d or 1) % R)
        inv = pow(pref[-1], -1, R)
        inv_d = [0] * (n - 1)
        for i in range(n - 2, -1, -1):
            inv_d[i] = pref[i] * inv % R if dens[i] else 0
            inv = inv * (dens[i] or 1) % R
        z = [1]
        for i in range(n - 1):
            z.append(z[-1] * nums[i] % R * inv_d[i] % R)
        z_ev = self.buf["z_ev"][:n]
        z_ev.copy_(_dev(_limbs(z)))
        return z_ev

    # ---- independent check of the commitments --------------------------------------------------------------------
    def committed_polynomials(self):
        """[(proof field, coefficient buffer, count)] of the LAST proof's nine commitments plus the eight of the preprocessing:
        with a known tau every one of them must equal p(tau) * G1 (tests, tools/bench_plonk.py)."""
        n, B = self.n, self.buf
        out = [("a_comm", B["w0"], n + 2), ("b_comm", B["w1"], n + 2), ("c_comm", B["w2"], n + 2), ("z_comm", B["z"], n + 3),
               ("t_lo_comm", B["t0"], n), ("t_mid_comm", B["t1"], n), ("t_hi_comm", B["t2"], n + 6),
               ("W_zeta_comm", B["q0"], n + 5), ("W_zeta_omega_comm", B["q1"], n + 2)]
        return out + [(k + "_comm", self.coef[k], n) for k in ("q_l", "q_r", "q_o", "q_m", "q_c", "s_sigma1", "s_sigma2", "s_sigma3")]

    def closed_form_mismatches(self, proof, tau, on_host=False, mul=None):
        """Names of the commitments that differ from p(tau) * G1.  p(tau) comes from the device (scale by the powers of tau,
        running sum: zk_fr_scale_powers_dev + zk_fr_scan_dev) or, on_host, from Horner's rule on Python integers over the
        downloaded coefficients -- no MSM, no SRS point involved either way.  mul(k) -> k * G1 as (x, y) integers | None lets the
        caller supply the scalar multiplication (tests and tools pass the C oracle's, so that the expected point owes nothing
        to this library); without it the library's own group operation is used."""
        from ..field import G1, ec_mul
        if mul is None:
            mul = lambda k: ec_mul(G1, k)
        as_ints = lambda pt: None if pt is None else (int(pt[0]), int(pt[1]))
        bad = []
        for name, coef, count in self.committed_polynomials():
            if on_host:
                val = 0
                for cv in reversed(_lib.limbs_to_ints(coef[:count].cpu().numpy().view(np.uint64))):
                    val = (val * tau + cv) % R
            else:
                val = int(self._evaluate(coef, count, tau))
            have = self.comm[name[:-5]] if name[:-5] in self.comm else getattr(proof, name)   # preprocessing / proof field
            if as_ints(have) != as_ints(mul(val)):
                bad.append(name)
        return bad

    # ---- interface to the verifier ---------------------------------------------------------------------------
    def preprocessed(self):
        """Object with the fields zkhip.plonk.verifier.verify reads (preprocessor.py:59-130)."""
        class PP:
            pass
        pp = PP()
        pp.n, pp.omega = self.n, self.omega
        for k, c in self.comm.items():
            setattr(pp, k + "_comm", c)
        return pp

    # ---- the five rounds ---------------------------------------------------------------------------------------
    def prove(self, a_vals, b_vals, c_vals, blinding=None, challenges=None):
        """a_vals, b_vals, c_vals: (n, 4) limb arrays or device tensors of the wire columns -> Proof.  blinding (nine scalars) and
        challenges ({"beta": .., "gamma": ..} in place of those two transcript values) exist for the tests that compare this
        prover with the oracle field by field; a caller leaves both None."""
        torch = _torch()
        n, size, step, fv, st = self.n, self.size, self.step, self.fv, self.st
        blind = list(blinding) if blinding is not None else [secrets.randbelow(R) for _ in range(9)]
        if len(blind) < 9:
            raise ValueError("not enough blinding scalars supplied")
        cols = [v if torch.is_tensor(v) else _dev(v) for v in (a_vals, b_vals, c_vals)]
        tr, pr = Transcript(), Proof()
        k1, k2 = int(K1), int(K2)

        # round 1 (round1.py:55-108)
        B = self.buf
        d_blind = _dev(_limbs(blind[:9]))                                             # all nine blinding scalars in one copy, before anything is queued
        wires = self._interpolate_many(cols, [B["w%d" % i] for i in range(3)])       # the three columns in one launch per pass
        wires = [self._blinded(w, d_blind[2 * i:2 * i + 2]) for i, w in enumerate(wires)]
        tickets = [self._submit(w, n + 2) for w in wires]
        ea, eb, ec = self._coset_many(wires, self.work[:3])   # round 3's coset evaluations of the wires need no challenge: under the MSMs
        for name, t in zip(("a_comm", "b_comm", "c_comm"), tickets):
            limbs, inf = self.msm.collect_limbs(t)
            comm = None if inf else limbs_to_g1(limbs)[0]
            setattr(pr, name, comm)
            tr.append_point(name.encode(), comm)

        # round 2 (round2.py:50-86, permutation.py:89-137)
        beta, gamma = tr.challenge_scalar(b"beta"), tr.challenge_scalar(b"gamma")
        if challenges:
            beta, gamma = FR(challenges.get("beta", int(beta))), FR(challenges.get("gamma", int(gamma)))
        be, ga = int(beta), int(gamma)
        z_ev = self._accumulator(cols, be, ga)
        z = self._interpolate(z_ev, B["z"])
        z = self._blinded(z, d_blind[6:9])
        t_z = self._submit(z, n + 3)
        ez = self._coset(z, self.work[3])                    # likewise under the commitment of z
        ezw = self.work[4]
        ezw[:size - step] = ez[step:]                        # z(omega x): omega = w_big^step
        ezw[size - step:] = ez[:step]
        limbs, inf = self.msm.collect_limbs(t_z)
        pr.z_comm = None if inf else limbs_to_g1(limbs)[0]
        tr.append_point(b"z_comm", pr.z_comm)

        # round 3 (round3.py:80-187): t = (gate + alpha perm + alpha^2 (z - 1) L1) / Z_H on the coset
        alpha = tr.challenge_scalar(b"alpha")
        al = int(alpha)
        cs = self.coset
        tot = self.work[5]
        plonk_quotient(tot.data_ptr(), [t.data_ptr() for t in (ea, eb, ec, ez, ezw, cs["q_l"], cs["q_r"], cs["q_o"], cs["q_m"], cs["q_c"],
                                                                 cs["s_sigma1"], cs["s_sigma2"], cs["s_sigma3"], cs["x"], cs["l1"])],
                       self.zh_inv, al, be, ga, size, st)                         # one pass over the 15 vectors (zk_plonk_quotient_dev)
        self.ntt_big.run(tot.data_ptr(), True, COSET_K, st)
        if bool(tot[3 * n + 6:].any()):
            raise ValueError(NOT_DIVISIBLE)
        t_parts = []
        for k, (lo, hi) in enumerate(((0, n), (n, 2 * n), (2 * n, 3 * n + 6))):
            part = B["t%d" % k]
            part[:hi - lo] = tot[lo:hi]
            part[hi - lo:].zero_()
            t_parts.append(part)
        for name, comm in zip(("t_lo_comm", "t_mid_comm", "t_hi_comm"), self._commit_many([(t_parts[0], n), (t_parts[1], n), (t_parts[2], n + 6)])):
            setattr(pr, name, comm)
            tr.append_point(name.encode(), comm)

        # round 4 (round4.py:40-79)
        zeta = tr.challenge_scalar(b"zeta")
        pr.a_eval, pr.b_eval, pr.c_eval, pr.s_sigma1_eval, pr.s_sigma2_eval = self._evaluate_many(
            [(wires[0], n + 2), (wires[1], n + 2), (wires[2], n + 2), (self.coef["s_sigma1"], n), (self.coef["s_sigma2"], n)], zeta)
        pr.z_omega_eval = self._evaluate(z, n + 3, zeta * self.omega)
        for name in ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval"):
            tr.append_scalar(name.encode(), getattr(pr, name))

        # round 5 (round5.py:78-177)
        v = tr.challenge_scalar(b"v")
        _, l1_zeta, perm_z, perm_s3, r0 = linearisation_scalars(alpha, beta, gamma, zeta, n, self.omega, pr.a_eval, pr.b_eval, pr.c_eval,
                                                               pr.s_sigma1_eval, pr.s_sigma2_eval, pr.z_omega_eval)
        cf = self.coef
        r_poly = B["r_poly"]
        self._lin(r_poly, [cf["q_m"], cf["q_l"], cf["q_r"], cf["q_o"], cf["q_c"], z, cf["s_sigma3"]],
                  [int(pr.a_eval * pr.b_eval), int(pr.a_eval), int(pr.b_eval), int(pr.c_eval), 1, int(perm_z + alpha * alpha * l1_zeta), int(FR(0) - perm_s3)],
                  n + PAD)
        self._lin(r_poly[:1], [r_poly[:1]], [1], 1, int(r0))                     # + (PI(zeta) + r0), PI = 0
        pr.r_eval = self._evaluate(r_poly, n + 3, zeta)
        zeta_n = zeta ** n
        numer = B["numer"]
        vs = [int(v ** k) for k in range(1, 7)]
        self._lin(numer, [t_parts[0], t_parts[1], t_parts[2], r_poly, wires[0], wires[1], wires[2], cf["s_sigma1"]],
                  [1, int(zeta_n), int(zeta_n * zeta_n), vs[0], vs[1], vs[2], vs[3], vs[4]], n + PAD)
        self._lin(numer, [numer, cf["s_sigma2"]], [1, vs[5]], n + PAD)
        w_zeta = self._divide_linear(numer, n + 6, zeta, B["q0"])                # the constant terms only change the dropped remainder
        w_zeta_omega = self._divide_linear(z, n + 3, zeta * self.omega, B["q1"])
        pr.W_zeta_comm, pr.W_zeta_omega_comm = self._commit_many([(w_zeta, n + 5), (w_zeta_omega, n + 2)])
        return pr
