"""
oracle/plonk_ref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Plain-int CPU restatement of the reference's PLONK flow IN ITS OWN SHAPE: coefficient-form
polynomials with the O(n^2) schoolbook product and long division, Horner evaluation, the
5-round prover with its Fiat-Shamir transcript, the preprocessor and the verifier.  Each
function cites the reference lines it follows (tokamak-network/interactive-zkp-study):

  zkp/plonk/polynomial.py:66-72,85-106,108-162,385-475  Polynomial._trim / evaluate / + - * / poly_div / lagrange_basis
  zkp/plonk/transcript.py:36-123                        Transcript
  zkp/plonk/circuit.py:157-331                          Gate selectors, build_copy_constraints, the x^3+x+5=35 circuit
  zkp/plonk/permutation.py:36-137                       K1, K2, build_permutation_polynomials, compute_accumulator
  zkp/plonk/preprocessor.py:59-130                      preprocess
  zkp/plonk/prover/round1.py:55-108 ... round5.py:78-175, prover/__init__.py:158-211
  zkp/plonk/verifier.py:42-208
  zkp/plonk/utils.py:25-80                              vanishing_poly_eval, lagrange_basis_eval

The curve / pairing / NTT / commit primitives come from oracle/py_ref.py (same status: the
arithmetic of py-ecc 7.0.1 restated; EC coordinates "parity unpinned", see its header).  The
product computes the round-3 quotient by coset NTTs and its commitments by Pippenger MSMs; this
file computes them the reference's way, so agreement of all 16 proof fields under injected blinding
pins transcript order, blinding layout, quotient split and linearisation to the reference's
definitions rather than to the product's own verifier.

Randomness: the reference draws its 9 blinding scalars with secrets.randbelow
(round1.py:106, round2.py:77); here they are an explicit argument, in draw order
(a: b1 b2, b: b1 b2, c: b1 b2, z: b1 b2 b3).

Representation: F_r elements are ints in [0, R); a polynomial is a list of coefficients, lowest
degree first, trimmed like Polynomial (at least one coefficient); points as in py_ref.
"""
import hashlib

import py_ref as pr

R = pr.R
K1, K2 = 2, 3                                  # permutation.py:36-37


# ----------------------------------------------------------------------------- Polynomial (coefficient form)
def P(coeffs):
    """Polynomial(coeffs): reduce mod r, trim trailing zeros, keep one coefficient (polynomial.py:52-72)."""
    return pr.trim(list(coeffs))


def p_add(a, b):
    """polynomial.py:108-118."""
    n = max(len(a), len(b))
    return P([(a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0) for i in range(n)])


def p_sub(a, b):
    """polynomial.py:123-133."""
    n = max(len(a), len(b))
    return P([(a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0) for i in range(n)])


def p_scale(a, k):
    """Polynomial * scalar (polynomial.py:150-153)."""
    return P([c * k for c in a])


def p_mul(a, b):
    """Polynomial * Polynomial: the O(n^2) convolution of polynomial.py:155-159."""
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % R
    return P(out)


def p_eval(a, x):
    """Horner (polynomial.py:85-106)."""
    return pr.horner(a, x % R)


def p_is_zero(a):
    return len(a) == 1 and a[0] == 0


def poly_div(a, b):
    """Long division a = b*q + rem (polynomial.py:385-435) -> (q, rem)."""
    if p_is_zero(b):
        raise ValueError("division by the zero polynomial")
    rem = list(a)
    deg_b, deg_a = len(b) - 1, len(rem) - 1
    if deg_a < deg_b:
        return P([0]), P(rem)
    quo = [0] * (deg_a - deg_b + 1)
    lead_inv = pr.fr_inv(b[-1])
    for i in range(deg_a - deg_b, -1, -1):
        coeff = rem[i + deg_b] * lead_inv % R
        quo[i] = coeff
        for j in range(deg_b + 1):
            rem[i + j] = (rem[i + j] - coeff * b[j]) % R
    return P(quo), P(rem)


def vanishing(n):
    """Z_H(x) = x^n - 1 (polynomial.py:243-276)."""
    return P([R - 1] + [0] * (n - 1) + [1])


def lagrange_basis(domain, i):
    """L_i(x) = prod_{j != i} (x - d_j) / (d_i - d_j) in coefficient form (polynomial.py:438-475)."""
    res, den = [1], 1
    for j, dj in enumerate(domain):
        if j == i:
            continue
        res = p_mul(res, P([-dj, 1]))
        den = den * (domain[i] - dj) % R
    return p_scale(res, pr.fr_inv(den))


def from_evaluations(evals, omega):
    """Polynomial.from_evaluations = Polynomial(ifft(evals, omega)) (polynomial.py:278-300)."""
    return P(pr.ifft([e % R for e in evals], omega))


def vanishing_poly_eval(n, zeta):
    """utils.py:25-43."""
    return (pow(zeta, n, R) - 1) % R


def lagrange_basis_eval(i, n, omega, zeta):
    """utils.py:46-80: L_i(zeta) = (omega^i / n) (zeta^n - 1) / (zeta - omega^i); 1 when zeta = omega^i."""
    omega_i = pow(omega, i, R)
    den = (zeta - omega_i) % R
    if den == 0:
        return 1
    return pr.fr_inv(n) * vanishing_poly_eval(n, zeta) % R * omega_i % R * pr.fr_inv(den) % R


# ----------------------------------------------------------------------------- transcript
class Transcript:
    """transcript.py:36-123."""

    def __init__(self, label=b"plonk"):
        self.state = bytearray()
        self.state.extend(label)

    def append_scalar(self, label, scalar):
        self.state.extend(label)
        self.state.extend((int(scalar) % R).to_bytes(32, "big"))

    def append_point(self, label, point):
        self.state.extend(label)
        if point is None:
            self.state.extend(b"\x00" * 64)
        else:
            self.state.extend(int(point[0]).to_bytes(32, "big"))
            self.state.extend(int(point[1]).to_bytes(32, "big"))

    def challenge_scalar(self, label):
        self.state.extend(label)
        h = hashlib.sha256(bytes(self.state)).digest()
        self.state.extend(h)
        return int.from_bytes(h, "big") % R


# ----------------------------------------------------------------------------- circuit
class Circuit:
    """Gates as (q_l, q_r, q_o, q_m, q_c) tuples, copy constraints as (g1, w1, g2, w2) (circuit.py:97-236)."""

    def __init__(self):
        self.gates = []
        self.copy_constraints = []
        self.num_public_inputs = 0

    @property
    def n(self):
        return len(self.gates)

    def add_multiplication_gate(self):      # circuit.py:116-128: a*b = c
        self.gates.append((0, 0, R - 1, 1, 0))
        return len(self.gates) - 1

    def add_addition_gate(self):            # circuit.py:130-142: a+b = c
        self.gates.append((1, 1, R - 1, 0, 0))
        return len(self.gates) - 1

    def add_constant_gate(self, constant):  # circuit.py:144-161: a+const = c
        self.gates.append((1, 0, R - 1, 0, constant % R))
        return len(self.gates) - 1

    def add_copy_constraint(self, g1, w1, g2, w2):
        self.copy_constraints.append((g1, w1, g2, w2))

    def get_selector_polynomials(self):     # circuit.py:198-211
        return tuple([g[k] for g in self.gates] for k in range(5))

    def build_copy_constraints(self):       # circuit.py:213-236: identity, then swap the images of the two positions
        n = self.n
        sigma = list(range(3 * n))
        for g1, w1, g2, w2 in self.copy_constraints:
            p1, p2 = w1 * n + g1, w2 * n + g2
            sigma[p1], sigma[p2] = sigma[p2], sigma[p1]
        return sigma


def circuit_x3_plus_x_plus_5_eq_35():
    """circuit.py:286-331 -> (circuit, a_vals, b_vals, c_vals, public_inputs); x = 3."""
    c = Circuit()
    c.add_multiplication_gate()
    c.add_multiplication_gate()
    c.add_addition_gate()
    c.add_constant_gate(5)
    for cc in ((0, 0, 0, 1), (0, 0, 1, 1), (0, 0, 2, 1), (0, 2, 1, 0), (1, 2, 2, 0), (2, 2, 3, 0)):
        c.add_copy_constraint(*cc)
    x = 3
    x2, x3 = x * x, x * x * x
    s = x3 + x
    c.num_public_inputs = 1
    return c, [x, x2, x3, s], [x, x, x, 0], [x2, x3, s, s + 5], [35]


def circuit_six_gates(x=5):
    """A second circuit (no reference counterpart; built with the reference's gate API): six gates, padded to n = 8 by
    preprocess.  ((x*x + x) * (x + 7) + x*x) * x with every shared value tied by a copy constraint.
    -> (circuit, a_vals, b_vals, c_vals, public_inputs) with the witness columns already padded to 8 rows of zeros,
    as a caller of the reference must do (from_evaluations needs n values)."""
    c = Circuit()
    c.add_multiplication_gate()     # 0: x * x      = x2
    c.add_addition_gate()           # 1: x2 + x     = u
    c.add_constant_gate(7)          # 2: x + 7      = v
    c.add_multiplication_gate()     # 3: u * v      = p
    c.add_addition_gate()           # 4: p + x2     = q
    c.add_multiplication_gate()     # 5: q * x      = out
    for cc in ((0, 0, 0, 1), (0, 0, 1, 1), (0, 0, 2, 0), (0, 0, 5, 1),      # x
               (0, 2, 1, 0), (0, 2, 4, 1),                                  # x2
               (1, 2, 3, 0), (2, 2, 3, 1), (3, 2, 4, 0), (4, 2, 5, 0)):     # u, v, p, q
        c.add_copy_constraint(*cc)
    x %= R
    x2 = x * x % R
    u, v = (x2 + x) % R, (x + 7) % R
    p = u * v % R
    q = (p + x2) % R
    out = q * x % R
    a = [x, x2, x, u, p, q, 0, 0]
    b = [x, x, 0, v, x2, x, 0, 0]
    cv = [x2, u, v, p, q, out, 0, 0]
    return c, a, b, cv, [out]


def gates_satisfied(circuit, a_vals, b_vals, c_vals):
    """Gate.check for every row (circuit.py:72-94)."""
    return all((g[0] * a + g[1] * b + g[2] * c + g[3] * a * b + g[4]) % R == 0
               for g, a, b, c in zip(circuit.gates, a_vals, b_vals, c_vals))


# ----------------------------------------------------------------------------- permutation
def build_permutation_polynomials(sigma, n, domain):
    """permutation.py:40-86 -> three evaluation vectors."""
    def value(pos):
        if pos < n:
            return domain[pos]
        if pos < 2 * n:
            return K1 * domain[pos - n] % R
        return K2 * domain[pos - 2 * n] % R
    return ([value(sigma[i]) for i in range(n)], [value(sigma[n + i]) for i in range(n)], [value(sigma[2 * n + i]) for i in range(n)])


def compute_accumulator(a_vals, b_vals, c_vals, sigma, n, domain, beta, gamma):
    """permutation.py:89-137: z_0 = 1, z_{i+1} = z_i * num_i / den_i (one field division per step, as the reference)."""
    s1, s2, s3 = build_permutation_polynomials(sigma, n, domain)
    z = [1]
    for i in range(n - 1):
        num = ((a_vals[i] + beta * domain[i] + gamma) * (b_vals[i] + beta * K1 * domain[i] + gamma) % R
               * (c_vals[i] + beta * K2 * domain[i] + gamma) % R)
        den = ((a_vals[i] + beta * s1[i] + gamma) * (b_vals[i] + beta * s2[i] + gamma) % R
               * (c_vals[i] + beta * s3[i] + gamma) % R)
        z.append(z[-1] * num % R * pr.fr_inv(den) % R)
    return z


# ----------------------------------------------------------------------------- preprocessing
class Preprocessed:
    pass


def next_power_of_2(n):
    p = 1
    while p < n:
        p <<= 1
    return p


def commit(poly, srs):
    """kzg.commit(poly, srs) with srs = (g1_powers, g2_powers) (kzg.py:32-67)."""
    return pr.kzg_commit(poly, srs[0])


def preprocess(circuit, srs):
    """preprocessor.py:59-130 (pads circuit.gates in place with all-zero gates)."""
    pp = Preprocessed()
    n = next_power_of_2(circuit.n)
    while len(circuit.gates) < n:
        circuit.gates.append((0, 0, 0, 0, 0))
    pp.n = n
    pp.omega = pr.get_root_of_unity(n)
    pp.domain = pr.get_roots_of_unity(n)
    for name, evals in zip(("q_l", "q_r", "q_o", "q_m", "q_c"), circuit.get_selector_polynomials()):
        poly = from_evaluations(evals, pp.omega)
        setattr(pp, name + "_poly", poly)
        setattr(pp, name + "_comm", commit(poly, srs))
    pp.sigma = circuit.build_copy_constraints()
    for k, evals in enumerate(build_permutation_polynomials(pp.sigma, n, pp.domain), start=1):
        poly = from_evaluations(evals, pp.omega)
        setattr(pp, "s_sigma%d_poly" % k, poly)
        setattr(pp, "s_sigma%d_comm" % k, commit(poly, srs))
    pp.num_public_inputs = circuit.num_public_inputs
    return pp


# ----------------------------------------------------------------------------- prover
PROOF_POINTS = ("a_comm", "b_comm", "c_comm", "z_comm", "t_lo_comm", "t_mid_comm", "t_hi_comm", "W_zeta_comm", "W_zeta_omega_comm")
PROOF_SCALARS = ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval", "r_eval")
PROOF_FIELDS = PROOF_POINTS[:7] + PROOF_SCALARS + PROOF_POINTS[7:]


class Proof:
    """prover/__init__.py:45-95."""

    def __init__(self):
        for f in PROOF_FIELDS:
            setattr(self, f, None)


class State:
    pass


def round1(st):
    """round1.py:55-108: PI = 0, interpolate a b c, blind with (b1 + b2 x) Z_H, commit, append."""
    n, omega = st.n, st.omega
    st.pi_poly = P([0])
    zh = vanishing(n)
    polys = []
    for vals in (st.a_vals, st.b_vals, st.c_vals):
        poly = from_evaluations(vals, omega)
        blind = P([st.blinding.pop(0) for _ in range(2)])           # _add_blinding(poly, zh, 2)
        polys.append(p_add(poly, p_mul(blind, zh)))
    st.a_poly, st.b_poly, st.c_poly = polys
    st.proof.a_comm, st.proof.b_comm, st.proof.c_comm = (commit(p, st.srs) for p in polys)
    st.transcript.append_point(b"a_comm", st.proof.a_comm)
    st.transcript.append_point(b"b_comm", st.proof.b_comm)
    st.transcript.append_point(b"c_comm", st.proof.c_comm)


def round2(st):
    """round2.py:50-86."""
    st.beta = st.transcript.challenge_scalar(b"beta")
    st.gamma = st.transcript.challenge_scalar(b"gamma")
    if st.challenges:                                  # tests only: a chosen beta / gamma (the transcript has absorbed its own all the same)
        st.beta = st.challenges.get("beta", st.beta) % R
        st.gamma = st.challenges.get("gamma", st.gamma) % R
    n = st.n
    z_evals = compute_accumulator(st.a_vals, st.b_vals, st.c_vals, st.pp.sigma, n, st.pp.domain, st.beta, st.gamma)
    z_poly = from_evaluations(z_evals, st.omega)
    blind = P([st.blinding.pop(0) for _ in range(3)])
    st.z_poly = p_add(z_poly, p_mul(blind, vanishing(n)))
    st.proof.z_comm = commit(st.z_poly, st.srs)
    st.transcript.append_point(b"z_comm", st.proof.z_comm)


def round3(st):
    """round3.py:64-184: the constraint polynomial by coefficient products, t = C / Z_H by long division, split in three."""
    st.alpha = st.transcript.challenge_scalar(b"alpha")
    n, omega, pp = st.n, st.omega, st.pp
    alpha, beta, gamma = st.alpha, st.beta, st.gamma
    a, b, c, z, pi = st.a_poly, st.b_poly, st.c_poly, st.z_poly, st.pi_poly
    # z(omega x): c_i -> omega^i c_i (round3.py:99-108)
    zw, wp = [], 1
    for coeff in z:
        zw.append(coeff * wp % R)
        wp = wp * omega % R
    z_omega = P(zw)
    x_poly = P([0, 1])
    l1 = lagrange_basis(pp.domain, 0)
    g = P([gamma])
    term1 = p_add(p_add(p_add(p_add(p_add(p_mul(pp.q_l_poly, a), p_mul(pp.q_r_poly, b)), p_mul(pp.q_o_poly, c)),
                              p_mul(pp.q_m_poly, p_mul(a, b))), pp.q_c_poly), pi)
    perm_num = p_mul(p_mul(p_mul(p_add(p_add(a, p_scale(x_poly, beta)), g),
                                 p_add(p_add(b, p_scale(x_poly, beta * K1 % R)), g)),
                           p_add(p_add(c, p_scale(x_poly, beta * K2 % R)), g)), z)
    perm_den = p_mul(p_mul(p_mul(p_add(p_add(a, p_scale(pp.s_sigma1_poly, beta)), g),
                                 p_add(p_add(b, p_scale(pp.s_sigma2_poly, beta)), g)),
                           p_add(p_add(c, p_scale(pp.s_sigma3_poly, beta)), g)), z_omega)
    term2 = p_scale(p_sub(perm_num, perm_den), alpha)
    term3 = p_scale(p_mul(p_sub(z, P([1])), l1), alpha * alpha % R)
    constraint = p_add(p_add(term1, term2), term3)
    t_poly, rem = poly_div(constraint, vanishing(n))
    if any(rem):
        raise ValueError("the constraint polynomial is not divisible by Z_H")
    t = list(t_poly)
    while len(t) < 3 * n:
        t.append(0)
    st.t_lo_poly, st.t_mid_poly = P(t[:n]), P(t[n:2 * n])
    st.t_hi_poly = P(t[2 * n:])                        # round3.py:158-164: coefficients beyond 3n stay in t_hi
    st.proof.t_lo_comm, st.proof.t_mid_comm, st.proof.t_hi_comm = (commit(p, st.srs) for p in (st.t_lo_poly, st.t_mid_poly, st.t_hi_poly))
    st.transcript.append_point(b"t_lo_comm", st.proof.t_lo_comm)
    st.transcript.append_point(b"t_mid_comm", st.proof.t_mid_comm)
    st.transcript.append_point(b"t_hi_comm", st.proof.t_hi_comm)


def round4(st):
    """round4.py:40-79."""
    st.zeta = st.transcript.challenge_scalar(b"zeta")
    zeta, pp, pf = st.zeta, st.pp, st.proof
    pf.a_eval = p_eval(st.a_poly, zeta)
    pf.b_eval = p_eval(st.b_poly, zeta)
    pf.c_eval = p_eval(st.c_poly, zeta)
    pf.s_sigma1_eval = p_eval(pp.s_sigma1_poly, zeta)
    pf.s_sigma2_eval = p_eval(pp.s_sigma2_poly, zeta)
    pf.z_omega_eval = p_eval(st.z_poly, zeta * st.omega % R)
    for name in ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval"):
        st.transcript.append_scalar(name.encode(), getattr(pf, name))


def round5(st):
    """round5.py:46-175."""
    st.v = st.transcript.challenge_scalar(b"v")
    v, n, zeta, omega, pp, pf = st.v, st.n, st.zeta, st.omega, st.pp, st.proof
    alpha, beta, gamma = st.alpha, st.beta, st.gamma
    a_e, b_e, c_e, s1_e, s2_e, zw_e = pf.a_eval, pf.b_eval, pf.c_eval, pf.s_sigma1_eval, pf.s_sigma2_eval, pf.z_omega_eval
    pi_zeta = p_eval(st.pi_poly, zeta)
    l1_zeta = lagrange_basis_eval(0, n, omega, zeta)
    r_poly = p_add(p_add(p_add(p_add(p_add(p_scale(pp.q_m_poly, a_e * b_e % R), p_scale(pp.q_l_poly, a_e)), p_scale(pp.q_r_poly, b_e)),
                               p_scale(pp.q_o_poly, c_e)), pp.q_c_poly), P([pi_zeta]))
    perm_z_scalar = alpha * (a_e + beta * zeta + gamma) % R * (b_e + beta * K1 * zeta + gamma) % R * (c_e + beta * K2 * zeta + gamma) % R
    ab_factor = (a_e + beta * s1_e + gamma) * (b_e + beta * s2_e + gamma) % R
    perm_s3_scalar = alpha * ab_factor % R * beta % R * zw_e % R
    perm_const = (-(alpha * ab_factor % R * zw_e % R * (c_e + gamma))) % R
    r_poly = p_add(r_poly, p_scale(st.z_poly, perm_z_scalar))
    r_poly = p_sub(r_poly, p_scale(pp.s_sigma3_poly, perm_s3_scalar))
    r_poly = p_add(r_poly, P([perm_const]))
    r_poly = p_add(r_poly, p_scale(st.z_poly, alpha * alpha % R * l1_zeta % R))
    r_poly = p_add(r_poly, P([-(alpha * alpha % R * l1_zeta)]))
    r_eval = p_eval(r_poly, zeta)
    pf.r_eval = r_eval
    zeta_n = pow(zeta, n, R)
    zeta_2n = zeta_n * zeta_n % R
    t_eval = (p_eval(st.t_lo_poly, zeta) + zeta_n * p_eval(st.t_mid_poly, zeta) + zeta_2n * p_eval(st.t_hi_poly, zeta)) % R
    t_combined = p_add(p_add(st.t_lo_poly, p_scale(st.t_mid_poly, zeta_n)), p_scale(st.t_hi_poly, zeta_2n))
    num = p_sub(t_combined, P([t_eval]))
    num = p_add(num, p_scale(p_sub(r_poly, P([r_eval])), v))
    v_pow = v * v % R
    for poly, val in ((st.a_poly, a_e), (st.b_poly, b_e), (st.c_poly, c_e), (pp.s_sigma1_poly, s1_e), (pp.s_sigma2_poly, s2_e)):
        num = p_add(num, p_scale(p_sub(poly, P([val])), v_pow))
        v_pow = v_pow * v % R
    w_zeta, st.rem_zeta = poly_div(num, P([-zeta, 1]))
    w_zeta_omega, st.rem_zeta_omega = poly_div(p_sub(st.z_poly, P([zw_e])), P([-(zeta * omega), 1]))
    st.r_poly, st.t_eval = r_poly, t_eval
    pf.W_zeta_comm = commit(w_zeta, st.srs)
    pf.W_zeta_omega_comm = commit(w_zeta_omega, st.srs)


def prove(circuit, a_vals, b_vals, c_vals, public_inputs, preprocessed, srs, blinding, return_state=False, challenges=None):
    """prover/__init__.py:158-211 with the 9 blinding scalars injected (draw order: a a b b c c z z z).  challenges: {"beta": ..,
    "gamma": ..} replaces those two Fiat-Shamir values (tests of the grand product's zero-denominator rows; the reference draws
    them from the transcript only)."""
    st = State()
    st.challenges = challenges
    st.a_vals, st.b_vals, st.c_vals = ([v % R for v in col] for col in (a_vals, b_vals, c_vals))
    st.public_inputs, st.pp, st.srs = public_inputs, preprocessed, srs
    st.transcript = Transcript()
    st.n, st.omega = preprocessed.n, preprocessed.omega
    st.blinding = [b % R for b in blinding]
    if len(st.blinding) != 9:
        raise ValueError("nine blinding scalars expected")
    st.proof = Proof()
    for rnd in (round1, round2, round3, round4, round5):
        rnd(st)
    return (st.proof, st) if return_state else st.proof


# ----------------------------------------------------------------------------- verifier
def verify(proof, public_inputs, preprocessed, srs):
    """verifier.py:42-208, step by step (one ec_mul / ec_add per term, two pairings)."""
    pp, pf = preprocessed, proof
    n, omega = pp.n, pp.omega
    ec_mul, ec_add, ec_neg = pr.ec_mul, pr.ec_add, pr.ec_neg
    tr = Transcript()
    tr.append_point(b"a_comm", pf.a_comm)
    tr.append_point(b"b_comm", pf.b_comm)
    tr.append_point(b"c_comm", pf.c_comm)
    beta = tr.challenge_scalar(b"beta")
    gamma = tr.challenge_scalar(b"gamma")
    tr.append_point(b"z_comm", pf.z_comm)
    alpha = tr.challenge_scalar(b"alpha")
    tr.append_point(b"t_lo_comm", pf.t_lo_comm)
    tr.append_point(b"t_mid_comm", pf.t_mid_comm)
    tr.append_point(b"t_hi_comm", pf.t_hi_comm)
    zeta = tr.challenge_scalar(b"zeta")
    for name in ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval"):
        tr.append_scalar(name.encode(), getattr(pf, name))
    v = tr.challenge_scalar(b"v")
    u = tr.challenge_scalar(b"u")
    a_e, b_e, c_e = pf.a_eval % R, pf.b_eval % R, pf.c_eval % R
    s1_e, s2_e, zw_e = pf.s_sigma1_eval % R, pf.s_sigma2_eval % R, pf.z_omega_eval % R
    zh_zeta = vanishing_poly_eval(n, zeta)
    l1_zeta = lagrange_basis_eval(0, n, omega, zeta)
    pi_zeta = 0
    D = ec_mul(pp.q_m_comm, a_e * b_e % R)
    D = ec_add(D, ec_mul(pp.q_l_comm, a_e))
    D = ec_add(D, ec_mul(pp.q_r_comm, b_e))
    D = ec_add(D, ec_mul(pp.q_o_comm, c_e))
    D = ec_add(D, pp.q_c_comm)
    perm_z_scalar = alpha * (a_e + beta * zeta + gamma) % R * (b_e + beta * K1 * zeta + gamma) % R * (c_e + beta * K2 * zeta + gamma) % R
    D = ec_add(D, ec_mul(pf.z_comm, perm_z_scalar))
    ab_factor = (a_e + beta * s1_e + gamma) * (b_e + beta * s2_e + gamma) % R
    perm_s3_scalar = alpha * ab_factor % R * beta % R * zw_e % R
    D = ec_add(D, ec_neg(ec_mul(pp.s_sigma3_comm, perm_s3_scalar)))
    D = ec_add(D, ec_mul(pf.z_comm, alpha * alpha % R * l1_zeta % R))
    r_0 = (pi_zeta - alpha * ab_factor % R * zw_e % R * (c_e + gamma) - alpha * alpha % R * l1_zeta) % R
    zeta_n = pow(zeta, n, R)
    zeta_2n = zeta_n * zeta_n % R
    t_comm = ec_add(pf.t_lo_comm, ec_add(ec_mul(pf.t_mid_comm, zeta_n), ec_mul(pf.t_hi_comm, zeta_2n)))
    F = t_comm
    F = ec_add(F, ec_mul(D, v))
    F = ec_add(F, ec_mul(pr.G1, v * r_0 % R))
    v_pow = v * v % R
    for comm in (pf.a_comm, pf.b_comm, pf.c_comm, pp.s_sigma1_comm, pp.s_sigma2_comm):
        F = ec_add(F, ec_mul(comm, v_pow))
        v_pow = v_pow * v % R
    r_eval = pf.r_eval % R
    t_eval = r_eval * pr.fr_inv(zh_zeta) % R                     # verifier.py:164; FR division: inv0(0) = 0 if zeta lies on the domain
    e_scalar = (t_eval + v * r_eval) % R
    v_pow = v * v % R
    for val in (a_e, b_e, c_e, s1_e, s2_e):
        e_scalar = (e_scalar + v_pow * val) % R
        v_pow = v_pow * v % R
    e_scalar = (e_scalar + u * zw_e) % R
    E = ec_mul(pr.G1, e_scalar)
    A = ec_add(pf.W_zeta_comm, ec_mul(pf.W_zeta_omega_comm, u))
    B = ec_mul(pf.W_zeta_comm, zeta)
    B = ec_add(B, ec_mul(pf.W_zeta_omega_comm, u * zeta % R * omega % R))
    B = ec_add(B, F)
    B = ec_add(B, ec_mul(pf.z_comm, u))
    B = ec_add(B, ec_neg(E))
    lhs = pr.pairing(srs[1][1], A)
    rhs = pr.pairing(srs[1][0], B)
    return lhs == rhs
