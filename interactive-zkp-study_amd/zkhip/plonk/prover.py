"""PLONK prover on the GPU backend (mirrors zkp/plonk/prover/__init__.py:158-211 and round1..5.py).

Same five rounds, transcript labels, blinding degrees and proof fields as the reference.  The
polynomial side uses the backend instead of coefficient algebra:
  * interpolation of the wire / accumulator columns: inverse NTT (GPU);
  * round 3: the quotient t = C / Z_H is computed pointwise on the coset 5*H' of a domain H' with
    |H'| >= deg t + 1 (coset NTTs of the operand polynomials, one coset inverse NTT of the result)
    instead of O(n^2) polynomial products and long division (round3.py:114-147);
    z(omega x) needs no extra transform (it is the same evaluation vector rotated by |H'|/n);
  * every commitment: one G1 MSM (GPU).
`blinding` lets tests inject the 9 blinding scalars the reference draws with secrets.randbelow
(round1.py:106, round2.py:77); by default they are random, so two proofs of the same witness differ.
"""
import secrets

from ..field import FR, CURVE_ORDER as R
from .kzg import commit
from .permutation import K1, K2, compute_accumulator
from .polynomial import Polynomial, poly_div
from .transcript import Transcript
from .utils import coset_fft, coset_ifft
from ..field import get_root_of_unity

COSET_K = 5


class Proof:
    FIELDS = ("a_comm", "b_comm", "c_comm", "z_comm", "t_lo_comm", "t_mid_comm", "t_hi_comm", "a_eval", "b_eval", "c_eval",
              "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval", "r_eval", "W_zeta_comm", "W_zeta_omega_comm")

    def __init__(self):
        for f in self.FIELDS:
            setattr(self, f, None)


class ProverState:
    def __init__(self, a_vals, b_vals, c_vals, public_inputs, preprocessed, srs, blinding=None):
        self.a_vals, self.b_vals, self.c_vals = ([FR(v) for v in col] for col in (a_vals, b_vals, c_vals))
        self.public_inputs = public_inputs
        self.preprocessed = preprocessed
        self.srs = srs
        self.transcript = Transcript()
        self.n, self.omega, self.domain = preprocessed.n, preprocessed.omega, preprocessed.domain
        self.a_poly = self.b_poly = self.c_poly = self.z_poly = None
        self.t_lo_poly = self.t_mid_poly = self.t_hi_poly = None
        self.beta = self.gamma = self.alpha = self.zeta = self.v = None
        self.pi_poly = None
        self.proof = Proof()
        self._blinding = list(blinding) if blinding is not None else None

    def _blind(self, count):
        if self._blinding is not None:
            out, self._blinding = self._blinding[:count], self._blinding[count:]
            if len(out) != count:
                raise ValueError("not enough blinding scalars supplied")
            return [FR(v) for v in out]
        return [FR(secrets.randbelow(R)) for _ in range(count)]

    def build_proof(self):
        return self.proof


def _pad_rows(vals, n):
    vals = list(vals)
    return vals + [FR(0)] * (n - len(vals))


def _times_vanishing(blind, n):
    """blind(x) * (x^n - 1)."""
    coeffs = [FR(0)] * (n + len(blind))
    for i, b in enumerate(blind):
        coeffs[i] = coeffs[i] - b
        coeffs[n + i] = coeffs[n + i] + b
    return Polynomial(coeffs)


def round1(state):
    """Wire polynomials a, b, c with degree-1 blinding, committed (round1.py:55-108)."""
    n, omega = state.n, state.omega
    state.pi_poly = Polynomial.zero()               # PI(x) is hard-wired to zero in the reference (round1.py:59)
    polys = []
    for vals in (state.a_vals, state.b_vals, state.c_vals):
        p = Polynomial.from_evaluations(_pad_rows(vals, n), omega)
        polys.append(p + _times_vanishing(state._blind(2), n))
    state.a_poly, state.b_poly, state.c_poly = polys
    for name, p in zip(("a_comm", "b_comm", "c_comm"), polys):
        comm = commit(p, state.srs)
        setattr(state.proof, name, comm)
        state.transcript.append_point(name.encode(), comm)


def round2(state):
    """Permutation accumulator z with degree-2 blinding (round2.py:50-86)."""
    state.beta = state.transcript.challenge_scalar(b"beta")
    state.gamma = state.transcript.challenge_scalar(b"gamma")
    n = state.n
    z_evals = compute_accumulator(_pad_rows(state.a_vals, n), _pad_rows(state.b_vals, n), _pad_rows(state.c_vals, n),
                                  state.preprocessed.sigma, n, state.domain, state.beta, state.gamma)
    state.z_poly = Polynomial.from_evaluations(z_evals, state.omega) + _times_vanishing(state._blind(3), n)
    state.proof.z_comm = commit(state.z_poly, state.srs)
    state.transcript.append_point(b"z_comm", state.proof.z_comm)


def _coset_evals(poly, size, omega_big):
    coeffs = [int(c) for c in poly.coeffs]
    return [int(v) for v in coset_fft(coeffs + [0] * (size - len(coeffs)), omega_big, FR(COSET_K))]


def round3(state):
    """Quotient polynomial t(x), split into three degree-<n parts and committed (round3.py:80-187)."""
    state.alpha = state.transcript.challenge_scalar(b"alpha")
    n, pp = state.n, state.preprocessed
    alpha, beta, gamma = int(state.alpha), int(state.beta), int(state.gamma)
    size = 1
    while size < 3 * n + 6:                          # deg t <= 3n + 5
        size <<= 1
    w_big = get_root_of_unity(size)
    ev = {name: _coset_evals(p, size, w_big) for name, p in (
        ("a", state.a_poly), ("b", state.b_poly), ("c", state.c_poly), ("z", state.z_poly),
        ("ql", pp.q_l_poly), ("qr", pp.q_r_poly), ("qo", pp.q_o_poly), ("qm", pp.q_m_poly), ("qc", pp.q_c_poly),
        ("s1", pp.s_sigma1_poly), ("s2", pp.s_sigma2_poly), ("s3", pp.s_sigma3_poly), ("pi", state.pi_poly))}
    step = size // n                                  # omega = w_big^step, so z(omega x_i) = z_evals[i + step]
    xs, cur = [], COSET_K % R
    wb = int(w_big)
    for _ in range(size):
        xs.append(cur)
        cur = cur * wb % R
    zh = [(pow(x, n, R) - 1) % R for x in xs[:step]]  # x^n takes only `step` distinct values on the coset
    zh_inv = [pow(v, -1, R) for v in zh]
    n_inv = pow(n, -1, R)
    k1, k2 = int(K1), int(K2)
    t_evals = []
    for i, x in enumerate(xs):
        a, b, c, z = ev["a"][i], ev["b"][i], ev["c"][i], ev["z"][i]
        zw = ev["z"][(i + step) % size]
        gate = (ev["ql"][i] * a + ev["qr"][i] * b + ev["qo"][i] * c + ev["qm"][i] * a % R * b + ev["qc"][i] + ev["pi"][i]) % R
        num = (a + beta * x + gamma) * (b + beta * k1 % R * x + gamma) % R * (c + beta * k2 % R * x + gamma) % R * z % R
        den = (a + beta * ev["s1"][i] + gamma) * (b + beta * ev["s2"][i] + gamma) % R * (c + beta * ev["s3"][i] + gamma) % R * zw % R
        l1 = zh[i % step] * n_inv % R * pow((x - 1) % R, -1, R) % R          # L_1(x) = (x^n - 1) / (n (x - 1))
        total = (gate + alpha * (num - den) + alpha * alpha % R * (z - 1) % R * l1) % R
        t_evals.append(total * zh_inv[i % step] % R)
    t_coeffs = [int(v) for v in coset_ifft(t_evals, w_big, FR(COSET_K))]
    if any(t_coeffs[3 * n + 6:]):
        raise ValueError("constraint polynomial is not divisible by Z_H: circuit or witness is inconsistent")
    t_coeffs = t_coeffs[:max(3 * n, 3 * n + 6)]
    while len(t_coeffs) > 3 * n and t_coeffs[-1] == 0:
        t_coeffs.pop()
    t_coeffs += [0] * (3 * n - len(t_coeffs))
    state.t_lo_poly = Polynomial(t_coeffs[:n])
    state.t_mid_poly = Polynomial(t_coeffs[n:2 * n])
    state.t_hi_poly = Polynomial(t_coeffs[2 * n:])   # keeps the overflow coefficients beyond 3n (round3.py:162-164)
    for name, p in (("t_lo_comm", state.t_lo_poly), ("t_mid_comm", state.t_mid_poly), ("t_hi_comm", state.t_hi_poly)):
        comm = commit(p, state.srs)
        setattr(state.proof, name, comm)
        state.transcript.append_point(name.encode(), comm)


def round4(state):
    """Openings at zeta and zeta*omega (round4.py:40-79)."""
    state.zeta = state.transcript.challenge_scalar(b"zeta")
    zeta, pp, pr = state.zeta, state.preprocessed, state.proof
    pr.a_eval = state.a_poly.evaluate(zeta)
    pr.b_eval = state.b_poly.evaluate(zeta)
    pr.c_eval = state.c_poly.evaluate(zeta)
    pr.s_sigma1_eval = pp.s_sigma1_poly.evaluate(zeta)
    pr.s_sigma2_eval = pp.s_sigma2_poly.evaluate(zeta)
    pr.z_omega_eval = state.z_poly.evaluate(zeta * state.omega)
    for name in ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval"):
        state.transcript.append_scalar(name.encode(), getattr(pr, name))


def linearisation_scalars(alpha, beta, gamma, zeta, n, omega, a_eval, b_eval, c_eval, s1_eval, s2_eval, z_omega_eval):
    """Scalars shared by round 5 and the verifier (round5.py:92-132, verifier.py:103-131)."""
    zh_zeta = zeta ** n - FR(1)
    l1_zeta = FR(1) if zeta == FR(1) else zh_zeta / (FR(n) * (zeta - FR(1)))
    perm_z = alpha * (a_eval + beta * zeta + gamma) * (b_eval + beta * K1 * zeta + gamma) * (c_eval + beta * K2 * zeta + gamma)
    ab = (a_eval + beta * s1_eval + gamma) * (b_eval + beta * s2_eval + gamma)
    perm_s3 = alpha * ab * beta * z_omega_eval
    r0 = FR(0) - alpha * ab * z_omega_eval * (c_eval + gamma) - alpha * alpha * l1_zeta
    return zh_zeta, l1_zeta, perm_z, perm_s3, r0


def round5(state):
    """Linearisation polynomial r(x), its evaluation, and the two opening proofs (round5.py:78-177)."""
    state.v = state.transcript.challenge_scalar(b"v")
    v, n, zeta, pp, pr = state.v, state.n, state.zeta, state.preprocessed, state.proof
    _, l1_zeta, perm_z, perm_s3, r0 = linearisation_scalars(state.alpha, state.beta, state.gamma, zeta, n, state.omega, pr.a_eval,
                                                           pr.b_eval, pr.c_eval, pr.s_sigma1_eval, pr.s_sigma2_eval, pr.z_omega_eval)
    pi_zeta = state.pi_poly.evaluate(zeta)
    r_poly = (pp.q_m_poly * (pr.a_eval * pr.b_eval) + pp.q_l_poly * pr.a_eval + pp.q_r_poly * pr.b_eval + pp.q_o_poly * pr.c_eval
              + pp.q_c_poly + state.z_poly * (perm_z + state.alpha * state.alpha * l1_zeta) - pp.s_sigma3_poly * perm_s3
              + Polynomial([pi_zeta + r0]))
    pr.r_eval = r_poly.evaluate(zeta)
    zeta_n = zeta ** n
    t_combined = state.t_lo_poly + state.t_mid_poly * zeta_n + state.t_hi_poly * (zeta_n * zeta_n)
    numer = t_combined - Polynomial([t_combined.evaluate(zeta)])
    v_pow = v
    for poly, value in ((r_poly, pr.r_eval), (state.a_poly, pr.a_eval), (state.b_poly, pr.b_eval), (state.c_poly, pr.c_eval),
                        (pp.s_sigma1_poly, pr.s_sigma1_eval), (pp.s_sigma2_poly, pr.s_sigma2_eval)):
        numer = numer + (poly - Polynomial([value])) * v_pow
        v_pow = v_pow * v
    w_zeta, _ = poly_div(numer, Polynomial([FR(0) - zeta, FR(1)]))
    w_zeta_omega, _ = poly_div(state.z_poly - Polynomial([pr.z_omega_eval]), Polynomial([FR(0) - zeta * state.omega, FR(1)]))
    pr.W_zeta_comm = commit(w_zeta, state.srs)
    pr.W_zeta_omega_comm = commit(w_zeta_omega, state.srs)


def prove(circuit, a_vals, b_vals, c_vals, public_inputs, preprocessed, srs, blinding=None):
    """prove(circuit, a, b, c, public_inputs, preprocessed, srs) -> Proof  (prover/__init__.py:158-211).
    `public_inputs` is carried but unused, as in the reference (PI(x) = 0)."""
    state = ProverState(a_vals, b_vals, c_vals, public_inputs, preprocessed, srs, blinding)
    for rnd in (round1, round2, round3, round4, round5):
        rnd(state)
    return state.build_proof()
