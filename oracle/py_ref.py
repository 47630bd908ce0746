"""
oracle/py_ref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Pure-Python (plain int) CPU restatement of the reference hot path of
tokamak-network/interactive-zkp-study:

  * the arithmetic the reference delegates to the third-party package
    py-ecc==7.0.1 (requirements.txt:14; source NOT vendored under /root/reference and
    not installed here): affine short-Weierstrass add/double/multiply on BN254
    G1 (y^2 = x^3 + 3 over F_p) and on the twist G2 (y^2 = x^3 + 3/(9+i) over
    F_p^2, i^2 = -1), None = point at infinity, recursive double-and-add
    `multiply`.  Restated from py_ecc's published algorithm
    (py_ecc/bn128/bn128_curve.py: double/add/multiply/neg) and anchored on the
    reference's call sites (zkp/groth16/proving.py:12-15, zkp/plonk/field.py:72-115).
  * the reference's own formulas: proof_a/b/c (zkp/groth16/proving.py:23-75),
    sigma11..sigma22 (zkp/groth16/setup.py:15-69), hxr and friends
    (zkp/groth16/poly_utils.py:17-125), kzg.commit (zkp/plonk/kzg.py:32-67),
    SRS.generate (zkp/plonk/srs.py:50-87), fft/ifft (zkp/plonk/polynomial.py:292-378),
    coset_fft/coset_ifft (zkp/plonk/utils.py:145-205), get_root_of_unity
    (zkp/plonk/field.py:145-182).

PARITY PINNING: py_ecc cannot be imported here (ordinary ModuleNotFoundError, SURVEY.md
section 8c), and the reference's tests hold no expected EC coordinate or NTT output
vector.  This restatement is therefore pinned by (i) the F_r known-answers that survive
in reference comments (zkp/groth16/backend.py:355,363-367), (ii) the relational identities
the reference's tests assert (tests/groth16/test_setup.py, tests/plonk/test_crypto.py:113-191,
tests/plonk/test_foundation.py:486-540) and (iii) on-curve / group-order checks.
EC coordinates are "parity unpinned" against the real py_ecc; see DESIGN.md.  Second sources added in round 2
(tests/test_oracle.py): SymPy's EllipticCurve for G1, SymPy's FiniteExtension arithmetic under the textbook formulas for
G2 and for FQ12 products / inverses, the public constants 2*G1 (EIP-196) and omega_{2^28}.

Representation: F_p / F_r elements are Python ints in [0, modulus); F_p^2 elements are
2-tuples (c0, c1) = c0 + c1*i; G1 points are (x, y) int tuples; G2 points are
((x0,x1),(y0,y1)); infinity is None.
"""
import hashlib

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
CURVE_ORDER = R
G1 = (1, 2)
G2 = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


# ----------------------------------------------------------------------------- F_p / F_p^2
def fp_inv(a):
    return pow(a % P, -1, P)


def fp2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def fp2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def fp2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def fp2_smul(a, k):
    return ((a[0] * k) % P, (a[1] * k) % P)


def fp2_inv(a):
    d = fp_inv(a[0] * a[0] + a[1] * a[1])
    return ((a[0] * d) % P, (-a[1] * d) % P)


def fp2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


B1 = 3
B2 = fp2_mul((3, 0), fp2_inv((9, 1)))  # py_ecc: b2 = FQ2([3, 0]) / FQ2([9, 1])


# ----------------------------------------------------------------------------- G1 affine (py_ecc shape)
def g1_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B1) % P == 0


def g1_double(pt):
    if pt is None:
        return None
    x, y = pt
    m = 3 * x * x * fp_inv(2 * y) % P
    nx = (m * m - 2 * x) % P
    ny = (-m * nx + m * x - y) % P
    return (nx, ny)


def g1_add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    x1, y1 = p1
    x2, y2 = p2
    if x2 == x1 and y2 == y1:
        return g1_double(p1)
    if x2 == x1:
        return None
    m = (y2 - y1) * fp_inv(x2 - x1) % P
    nx = (m * m - x1 - x2) % P
    ny = (-m * nx + m * x1 - y1) % P
    return (nx, ny)


def g1_neg(pt):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def g1_multiply(pt, n):
    """py_ecc.bn128.multiply: recursive double-and-add, n a non-negative int."""
    if n == 0 or pt is None:
        return None
    if n == 1:
        return pt
    if not n % 2:
        return g1_multiply(g1_double(pt), n // 2)
    return g1_add(g1_multiply(g1_double(pt), n // 2), pt)


# ----------------------------------------------------------------------------- G2 affine
def g2_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    lhs = fp2_mul(y, y)
    rhs = fp2_add(fp2_mul(fp2_mul(x, x), x), B2)
    return lhs == rhs


def g2_double(pt):
    if pt is None:
        return None
    x, y = pt
    m = fp2_mul(fp2_smul(fp2_mul(x, x), 3), fp2_inv(fp2_smul(y, 2)))
    nx = fp2_sub(fp2_mul(m, m), fp2_smul(x, 2))
    ny = fp2_sub(fp2_sub(fp2_mul(m, x), fp2_mul(m, nx)), y)
    return (nx, ny)


def g2_add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    x1, y1 = p1
    x2, y2 = p2
    if x2 == x1 and y2 == y1:
        return g2_double(p1)
    if x2 == x1:
        return None
    m = fp2_mul(fp2_sub(y2, y1), fp2_inv(fp2_sub(x2, x1)))
    nx = fp2_sub(fp2_sub(fp2_mul(m, m), x1), x2)
    ny = fp2_sub(fp2_sub(fp2_mul(m, x1), fp2_mul(m, nx)), y1)
    return (nx, ny)


def g2_neg(pt):
    if pt is None:
        return None
    return (pt[0], fp2_neg(pt[1]))


def g2_multiply(pt, n):
    if n == 0 or pt is None:
        return None
    if n == 1:
        return pt
    if not n % 2:
        return g2_multiply(g2_double(pt), n // 2)
    return g2_add(g2_multiply(g2_double(pt), n // 2), pt)


def _is_g2(pt):
    return pt is not None and isinstance(pt[0], tuple)


def ec_add(p1, p2):
    """bn128.add dispatching on the point type, as the reference's `add` alias does."""
    if _is_g2(p1) or _is_g2(p2):
        return g2_add(p1, p2)
    return g1_add(p1, p2)


def ec_mul(pt, n):
    """zkp/plonk/field.py:72-88: scalar reduced mod r, then bn128.multiply."""
    n = int(n) % CURVE_ORDER
    return g2_multiply(pt, n) if _is_g2(pt) else g1_multiply(pt, n)


def ec_neg(pt):
    return g2_neg(pt) if _is_g2(pt) else g1_neg(pt)


# ----------------------------------------------------------------------------- F_r helpers
def fr_inv(a):
    """FR(1) / FR(a) as py_ecc computes it: prime_field_inv extends the inverse by inv0(0) = 0 (py_ecc/utils.py)."""
    a %= R
    return pow(a, -1, R) if a else 0


def get_root_of_unity(n):
    """zkp/plonk/field.py:145-182."""
    if n < 1 or (n & (n - 1)) != 0:
        raise ValueError("n must be a power of two: %d" % n)
    if n > (1 << 28):
        raise ValueError("n must be <= 2^28: %d" % n)
    if n == 1:
        return 1
    return pow(5, (R - 1) // n, R)


def get_roots_of_unity(n):
    w = get_root_of_unity(n)
    out, cur = [], 1
    for _ in range(n):
        out.append(cur)
        cur = cur * w % R
    return out


# ----------------------------------------------------------------------------- NTT (reference shape)
def fft(coeffs, omega):
    """zkp/plonk/polynomial.py:292-341: recursive radix-2 DIT, natural order in/out."""
    n = len(coeffs)
    if n == 1:
        return [coeffs[0] % R]
    even = [coeffs[i] for i in range(0, n, 2)]
    odd = [coeffs[i] for i in range(1, n, 2)]
    omega_sq = omega * omega % R
    ev = fft(even, omega_sq)
    od = fft(odd, omega_sq)
    res = [0] * n
    wk = 1
    half = n // 2
    for k in range(half):
        t = wk * od[k] % R
        res[k] = (ev[k] + t) % R
        res[k + half] = (ev[k] - t) % R
        wk = wk * omega % R
    return res


def ifft(evals, omega):
    """zkp/plonk/polynomial.py:344-378."""
    n = len(evals)
    omega_inv = fr_inv(omega)
    coeffs = fft(evals, omega_inv)
    n_inv = fr_inv(n)
    return [c * n_inv % R for c in coeffs]


def coset_fft(coeffs, omega, k=None):
    """zkp/plonk/utils.py:145-176."""
    if k is None:
        k = 5
    shifted, kp = [], 1
    for c in coeffs:
        shifted.append(c % R * kp % R)
        kp = kp * k % R
    return fft(shifted, omega)


def coset_ifft(evals, omega, k=None):
    """zkp/plonk/utils.py:179-205."""
    if k is None:
        k = 5
    coeffs = ifft(evals, omega)
    k_inv = fr_inv(k)
    kp, out = 1, []
    for c in coeffs:
        out.append(c * kp % R)
        kp = kp * k_inv % R
    return out


def horner(coeffs, x):
    """Polynomial.evaluate, zkp/plonk/polynomial.py:85-106."""
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


# ----------------------------------------------------------------------------- KZG / SRS
def srs_tau(seed):
    """zkp/plonk/srs.py:68-70."""
    h = hashlib.sha256(str(seed).encode()).digest()
    return int.from_bytes(h, "big") % CURVE_ORDER


def srs_generate(max_degree, seed):
    """zkp/plonk/srs.py:50-87 -> (g1_powers, g2_powers)."""
    tau = srs_tau(seed)
    g1_powers, tp = [], 1
    for _ in range(max_degree + 1):
        g1_powers.append(ec_mul(G1, tp))
        tp = tp * tau % R
    g2_powers = [G2, ec_mul(G2, tau)]
    return g1_powers, g2_powers


def trim(coeffs):
    """Polynomial._trim, zkp/plonk/polynomial.py:66-72."""
    c = [x % R for x in coeffs] or [0]
    while len(c) > 1 and c[-1] == 0:
        c.pop()
    return c


def kzg_commit(coeffs, g1_powers, max_degree=None):
    """zkp/plonk/kzg.py:32-67 (coeffs already trimmed as Polynomial does)."""
    coeffs = trim(coeffs)
    degree = 0 if coeffs == [0] else len(coeffs) - 1
    if max_degree is None:
        max_degree = len(g1_powers) - 1
    if degree > max_degree:
        raise ValueError("polynomial degree %d exceeds SRS max degree %d" % (degree, max_degree))
    result = None
    for i, c in enumerate(coeffs):
        if c == 0:
            continue
        result = ec_add(result, ec_mul(g1_powers[i], c))
    return result


def msm_naive(scalars, points):
    """The generic shape of kzg.commit / proof_*: sum_i s_i * P_i, affine double-and-add."""
    acc = None
    for s, pt in zip(scalars, points):
        s = int(s) % R
        if s == 0:
            continue
        acc = ec_add(acc, ec_mul(pt, s))
    return acc


# ----------------------------------------------------------------------------- Groth16 poly_utils
def multiply_polys(a, b):
    o = [0] * (len(a) + len(b) - 1)
    for i in range(len(a)):
        for j in range(len(b)):
            o[i + j] = (o[i + j] + a[i] * b[j]) % R
    return o


def subtract_polys(a, b):
    o = [0] * max(len(a), len(b))
    for i in range(len(a)):
        o[i] = (o[i] + a[i]) % R
    for i in range(len(b)):
        o[i] = (o[i] - b[i]) % R
    return o


def div_polys(a, b):
    """zkp/groth16/poly_utils.py:37-45."""
    o = [0] * (len(a) - len(b) + 1)
    rem = list(a)
    while len(rem) >= len(b):
        lead = rem[-1] * fr_inv(b[-1]) % R
        pos = len(rem) - len(b)
        o[pos] = lead
        rem = subtract_polys(rem, multiply_polys(b, [0] * pos + [lead]))[:-1]
    return o, rem


def eval_poly(poly, x):
    return sum(poly[i] * pow(x, i, R) for i in range(len(poly))) % R


def multiply_vec_matrix(vec, matrix):
    """zkp/groth16/poly_utils.py:52-59 (quirk: result has len(vec) entries, asserts W != G)."""
    assert not len(vec) == len(matrix[0])
    target = [0] * len(vec)
    for i in range(len(matrix)):
        for j in range(len(matrix[0])):
            target[j] = (target[j] + vec[i] * matrix[i][j]) % R
    return target


def hxr(Ax, Bx, Cx, Zx, Rv):
    """zkp/groth16/poly_utils.py:116-125."""
    Rax = multiply_vec_matrix(Rv, Ax)
    Rbx = multiply_vec_matrix(Rv, Bx)
    Rcx = multiply_vec_matrix(Rv, Cx)
    Px = subtract_polys(multiply_polys(Rax, Rbx), Rcx)
    return div_polys(Px, Zx)


# ----------------------------------------------------------------------------- Groth16 setup
PLACEHOLDER = (0, 0)  # (FQ(0), FQ(0)) in setup.py:39,50 -- not a curve point


def sigma11(alpha, beta, delta):
    return [g1_multiply(G1, alpha % R), g1_multiply(G1, beta % R), g1_multiply(G1, delta % R)]


def sigma12(numGates, x_val):
    return [g1_multiply(G1, pow(x_val, i, R)) for i in range(numGates)]


def sigma13(numWires, alpha, beta, gamma, Ax_val, Bx_val, Cx_val, pub=None):
    if pub is None:
        pub = [0, 1]
    out, VAL = [], [0] * numWires
    for i in range(numWires):
        if i in pub:
            val = (beta * Ax_val[i] + alpha * Bx_val[i] + Cx_val[i]) * fr_inv(gamma) % R
            VAL[i] = val
            out.append(g1_multiply(G1, val))
        else:
            out.append(PLACEHOLDER)
    return out, VAL


def sigma14(numWires, alpha, beta, delta, Ax_val, Bx_val, Cx_val, pub=None):
    if pub is None:
        pub = [0, 1]
    out = []
    for i in range(numWires):
        if i in pub:
            out.append(PLACEHOLDER)
        else:
            val = (beta * Ax_val[i] + alpha * Bx_val[i] + Cx_val[i]) * fr_inv(delta) % R
            out.append(g1_multiply(G1, val))
    return out


def sigma15(numGates, delta, x_val, Zx_val):
    return [g1_multiply(G1, pow(x_val, i, R) * Zx_val % R * fr_inv(delta) % R) for i in range(numGates - 1)]


def sigma21(beta, delta, gamma):
    return [g2_multiply(G2, beta % R), g2_multiply(G2, gamma % R), g2_multiply(G2, delta % R)]


def sigma22(numGates, x_val):
    return [g2_multiply(G2, pow(x_val, i, R)) for i in range(numGates)]


# ----------------------------------------------------------------------------- Groth16 proving
def _proof_lin(base0, bases, M, Rx, tail_pt, tail_s):
    """Shared nested-loop shape of proving.py:23-45,55-61."""
    numWires, numGates = len(M), len(M[0])
    acc = base0
    for i in range(numWires):
        temp = None
        for j in range(numGates):
            temp = ec_add(temp, ec_mul(bases[j], M[i][j]))
        acc = ec_add(acc, ec_mul(temp, Rx[i]))
    return ec_add(acc, ec_mul(tail_pt, tail_s))


def proof_a(sigma1_1, sigma1_2, Ax, Rx, r):
    return _proof_lin(sigma1_1[0], sigma1_2, Ax, Rx, sigma1_1[2], r)


def proof_b(sigma2_1, sigma2_2, Bx, Rx, s):
    return _proof_lin(sigma2_1[0], sigma2_2, Bx, Rx, sigma2_1[2], s)


def proof_c(sigma1_1, sigma1_2, sigma1_4, sigma1_5, Bx, Rx, Hx, s, r, prf_A, pub=None):
    if pub is None:
        pub = [0, 1]
    numWires, numGates = len(Bx), len(Bx[0])
    tB = _proof_lin(sigma1_1[1], sigma1_2, Bx, Rx, sigma1_1[2], s)
    C = ec_add(ec_add(ec_mul(prf_A, s), ec_mul(tB, r)),
               ec_neg(ec_mul(ec_mul(sigma1_1[2], s), r)))
    for i in range(numWires):
        if i in pub:
            continue
        C = ec_add(C, ec_mul(sigma1_4[i], Rx[i]))
    for i in range(numGates - 1):
        C = ec_add(C, ec_mul(sigma1_5[i], Hx[i]))
    return C


# ----------------------------------------------------------------------------- toy circuit (tests/groth16/conftest.py:39-56)
TOY = dict(
    R=[1, 3, 35, 9, 27, 30],
    alpha=3926, beta=3604, gamma=2971, delta=1357, x_val=3721,
    r=4106, s=4565, pub=[0, 1],
    # r1cs_to_qap_times_lcm output == zkp/groth16/backend.py:85-110 (integers as floats there)
    Ap=[[-60, 110, -60, 10], [96, -136, 60, -8], [0, 0, 0, 0],
        [-72, 114, -48, 6], [48, -84, 42, -6], [-12, 22, -12, 2]],
    Bp=[[36, -62, 30, -4], [-24, 62, -30, 4], [0, 0, 0, 0],
        [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]],
    Cp=[[0, 0, 0, 0], [0, 0, 0, 0], [-144, 264, -144, 24],
        [576, -624, 216, -24], [-864, 1368, -576, 72], [576, -1008, 504, -72]],
    Z=[24, -50, 35, -10, 1],   # live unscaled Z, qap_creator_lcm.py:129-133
)


def toy_groth16():
    """Runs the full_pipeline_data sequence (tests/groth16/conftest.py:82-158) on plain ints."""
    t = TOY
    fr = lambda v: v % R
    Ax = [[fr(v) for v in row] for row in t["Ap"]]
    Bx = [[fr(v) for v in row] for row in t["Bp"]]
    Cx = [[fr(v) for v in row] for row in t["Cp"]]
    Zx = [fr(v) for v in t["Z"]]
    Rx = [fr(v) for v in t["R"]]
    Hx, rem = hxr(Ax, Bx, Cx, Zx, Rx)
    numWires, numGates = len(Ax), len(Ax[0])
    x = t["x_val"]
    Axv = [eval_poly(p, x) for p in Ax]
    Bxv = [eval_poly(p, x) for p in Bx]
    Cxv = [eval_poly(p, x) for p in Cx]
    Zxv = eval_poly(Zx, x)
    s11 = sigma11(t["alpha"], t["beta"], t["delta"])
    s12 = sigma12(numGates, x)
    s13, VAL = sigma13(numWires, t["alpha"], t["beta"], t["gamma"], Axv, Bxv, Cxv, t["pub"])
    s14 = sigma14(numWires, t["alpha"], t["beta"], t["delta"], Axv, Bxv, Cxv, t["pub"])
    s15 = sigma15(numGates, t["delta"], x, Zxv)
    s21 = sigma21(t["beta"], t["delta"], t["gamma"])
    s22 = sigma22(numGates, x)
    pa = proof_a(s11, s12, Ax, Rx, t["r"])
    pb = proof_b(s21, s22, Bx, Rx, t["s"])
    pc = proof_c(s11, s12, s14, s15, Bx, Rx, Hx, t["s"], t["r"], pa, t["pub"])
    # closed-form scalars (zkp/groth16/test.py:303-325)
    A = (t["alpha"] + sum(Rx[i] * Axv[i] for i in range(numWires)) + t["r"] * t["delta"]) % R
    B = (t["beta"] + sum(Rx[i] * Bxv[i] for i in range(numWires)) + t["s"] * t["delta"]) % R
    Hxv = eval_poly(Hx, x)
    dinv = fr_inv(t["delta"])
    priv = sum(Rx[i] * (t["beta"] * Axv[i] + t["alpha"] * Bxv[i] + Cxv[i])
               for i in range(numWires) if i not in t["pub"]) % R
    C = (dinv * (priv + Hxv * Zxv) + A * t["s"] + B * t["r"] - t["r"] * t["s"] * t["delta"]) % R
    return dict(Ax=Ax, Bx=Bx, Cx=Cx, Zx=Zx, Rx=Rx, Hx=Hx, rem=rem, VAL=VAL,
                Ax_val=Axv, Bx_val=Bxv, Cx_val=Cxv, Zx_val=Zxv,
                s11=s11, s12=s12, s13=s13, s14=s14, s15=s15, s21=s21, s22=s22,
                proof_A=pa, proof_B=pb, proof_C=pc, A=A, B=B, C=C)


# ----------------------------------------------------------------------------- pairing (py_ecc shape)
# Restatement of py_ecc/bn128/bn128_pairing.py + bn128_curve.py (twist) + fields (FQ12 as
# F_p[w]/(w^12 - 18 w^6 + 82)), used by the reference through `pairing(Q, P)`:
# zkp/groth16/verifying.py:17-40, zkp/plonk/field.py:118-138, zkp/plonk/kzg.py:117-160.
FQ12_MOD = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]  # w^12 = 18 w^6 - 82
ATE_LOOP_COUNT = 29793968203157093288
LOG_ATE_LOOP_COUNT = 63


def f12(coeffs):
    return [c % P for c in coeffs]


F12_ONE = f12([1] + [0] * 11)
F12_ZERO = [0] * 12


def f12_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(22, 11, -1):  # w^k = 18 w^(k-6) - 82 w^(k-12)
        top = t[k]
        if top:
            t[k - 6] += 18 * top
            t[k - 12] -= 82 * top
    return [v % P for v in t[:12]]


def f12_smul(a, k):
    return [x * k % P for x in a]


def _poly_deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    """Extended Euclid in F_p[w] (py_ecc FQP.inv)."""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [v % P for v in FQ12_MOD] + [1]
    while _poly_deg(low):
        # r = high / low (polynomial division, quotient only)
        dega, degb = _poly_deg(high), _poly_deg(low)
        temp, o = list(high), [0] * 13
        inv_lead = pow(low[degb], -1, P)
        for i in range(dega - degb, -1, -1):
            q = temp[degb + i] * inv_lead % P
            o[i] = q
            for c in range(degb + 1):
                temp[c + i] = (temp[c + i] - low[c] * q) % P
        r = o
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                new[i + j] = (new[i + j] - low[i] * r[j]) % P
        lm, low, hm, high = nm, new, lm, low
    inv0 = pow(low[0], -1, P)
    return [v * inv0 % P for v in lm[:12]]


def f12_pow(a, e):
    r, base = list(F12_ONE), list(a)
    while e:
        if e & 1:
            r = f12_mul(r, base)
        base = f12_mul(base, base)
        e >>= 1
    return r


def twist(pt):
    """E'(F_p^2) -> E(F_p^12)  (py_ecc bn128_curve.twist)."""
    if pt is None:
        return None
    (x0, x1), (y0, y1) = pt
    nx = f12([x0 - 9 * x1] + [0] * 5 + [x1] + [0] * 5)
    ny = f12([y0 - 9 * y1] + [0] * 5 + [y1] + [0] * 5)
    w2 = f12([0, 0, 1] + [0] * 9)
    w3 = f12([0, 0, 0, 1] + [0] * 8)
    return (f12_mul(nx, w2), f12_mul(ny, w3))


def cast_g1_to_f12(pt):
    if pt is None:
        return None
    return (f12([pt[0]] + [0] * 11), f12([pt[1]] + [0] * 11))


def _f12_pt_double(pt):
    x, y = pt
    m = f12_mul(f12_smul(f12_mul(x, x), 3), f12_inv(f12_smul(y, 2)))
    nx = f12_sub(f12_mul(m, m), f12_smul(x, 2))
    ny = f12_sub(f12_sub(f12_mul(m, x), f12_mul(m, nx)), y)
    return (nx, ny)


def _f12_pt_add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    x1, y1 = p1
    x2, y2 = p2
    if x2 == x1 and y2 == y1:
        return _f12_pt_double(p1)
    if x2 == x1:
        return None
    m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    nx = f12_sub(f12_sub(f12_mul(m, m), x1), x2)
    ny = f12_sub(f12_sub(f12_mul(m, x1), f12_mul(m, nx)), y1)
    return (nx, ny)


def linefunc(P1, P2, T):
    """py_ecc bn128_pairing.linefunc on F_p^12 points."""
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(f12_smul(f12_mul(x1, x1), 3), f12_inv(f12_smul(y1, 2)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    return f12_sub(xt, x1)


def miller_loop(Q, Pt):
    if Q is None or Pt is None:
        return list(F12_ONE)
    R, f = Q, list(F12_ONE)
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f12_mul(f12_mul(f, f), linefunc(R, R, Pt))
        R = _f12_pt_double(R)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, linefunc(R, Q, Pt))
            R = _f12_pt_add(R, Q)
    Q1 = (f12_pow(Q[0], P), f12_pow(Q[1], P))
    nQ2 = (f12_pow(Q1[0], P), [(-v) % P for v in f12_pow(Q1[1], P)])
    f = f12_mul(f, linefunc(R, Q1, Pt))
    R = _f12_pt_add(R, Q1)
    f = f12_mul(f, linefunc(R, nQ2, Pt))
    return f12_pow(f, (P ** 12 - 1) // CURVE_ORDER)


def pairing(Q, Pt):
    """py_ecc.bn128.pairing(Q in G2, P in G1) -> F_p^12 coefficient list (12 ints).  Like py_ecc's, it asserts that
    both arguments lie on their curves (bn128_pairing.pairing: `assert is_on_curve(Q, b2)`, `assert is_on_curve(P, b)`;
    infinity passes)."""
    assert g2_is_on_curve(Q), "pairing: Q is not on the twist"
    assert g1_is_on_curve(Pt), "pairing: P is not on the curve"
    return miller_loop(twist(Q), cast_g1_to_f12(Pt))


def groth16_verify(prf_A, prf_B, prf_C, sigma1_1, sigma1_3, sigma2_1, rx_pub):
    """zkp/groth16/verifying.py:29-40."""
    lhs = pairing(prf_B, prf_A)
    rhs = pairing(sigma2_1[0], sigma1_1[0])
    temp = None
    for i, ri in rx_pub:
        temp = g1_add(temp, g1_multiply(sigma1_3[i], int(ri) % R))
    rhs = f12_mul(f12_mul(rhs, pairing(sigma2_1[1], temp)), pairing(sigma2_1[2], prf_C))
    return lhs == rhs
