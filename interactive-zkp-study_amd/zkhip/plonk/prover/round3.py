"""PLONK prover round 3 (mirrors zkp/plonk/prover/round3.py:64-184): challenge alpha; the quotient t = C / Z_H, split into three
parts of n coefficients and committed.  The reference builds C by O(n^2) coefficient products and divides by Z_H with poly_div
(round3.py:114-147); here t is computed pointwise on the coset 5*H' of a domain H' with |H'| >= deg t + 1 (coset NTTs of the operand
polynomials on the GPU, one coset inverse NTT of the result) -- the same polynomial, checked coefficient by
coefficient against the reference-shaped oracle (tests/test_gpu_plonk_golden.py).  z(omega x) needs no transform of its own: it is
the evaluation vector of z rotated by |H'|/n."""
from ...field import FR, CURVE_ORDER as R, get_root_of_unity
from ..kzg import commit
from ..permutation import K1, K2
from ..polynomial import Polynomial
from ..utils import coset_fft, coset_ifft
from .common import COSET_K, NOT_DIVISIBLE


def _coset_evals(poly, size, omega_big):
    coeffs = [int(c) for c in poly.coeffs]
    return [int(v) for v in coset_fft(coeffs + [0] * (size - len(coeffs)), omega_big, FR(COSET_K))]


def execute(state):
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
        raise ValueError(NOT_DIVISIBLE)
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
