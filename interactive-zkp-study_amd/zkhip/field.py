"""Host-side value types and the EC wrapper seam of the reference, backed by libzkhip.

Mirrors zkp/plonk/field.py of the reference (FR :36-51, CURVE_ORDER :55, G1/G2/Z1 :63-69,
ec_mul/ec_add/ec_neg :72-115, get_root_of_unity :145-182, get_roots_of_unity :185-209) and the
py_ecc value conventions the reference relies on:

  * FQ / FR: residue objects supporting int(), ==, + - * / ** and mixing with ints;
  * FQ2: `.coeffs` = (c0, c1) with i^2 = -1;
  * G1 point = (FQ, FQ), G2 point = (FQ2, FQ2), point at infinity = None.

Scalar glue (a handful of field operations per call) runs on Python ints; every group
operation goes through the C ABI to the GPU.
"""
import numpy as np

from . import _lib

FIELD_MODULUS = 21888242871839275222246405745257275088696311157297823662689037894645226208583
CURVE_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class _PrimeField:
    """Residue modulo `field_modulus` (py_ecc FQ semantics)."""
    field_modulus = None
    __slots__ = ("n",)

    def __init__(self, val):
        if isinstance(val, _PrimeField):
            val = val.n
        self.n = int(val) % self.field_modulus

    def _coerce(self, other):
        if isinstance(other, _PrimeField):
            return other.n
        return int(other)

    def __add__(self, other):
        return type(self)(self.n + self._coerce(other))

    __radd__ = __add__

    def __sub__(self, other):
        return type(self)(self.n - self._coerce(other))

    def __rsub__(self, other):
        return type(self)(self._coerce(other) - self.n)

    def __mul__(self, other):
        return type(self)(self.n * self._coerce(other))

    __rmul__ = __mul__

    @classmethod
    def _inv0(cls, v):
        """py_ecc's prime_field_inv: the modular inverse extended by inv0(0) = 0 -- the reference's FR(x) / FR(0) is FR(0),
        not an exception (py_ecc/utils.py, after draft-irtf-cfrg-hash-to-curve section 4)."""
        v %= cls.field_modulus
        return pow(v, -1, cls.field_modulus) if v else 0

    def __truediv__(self, other):
        return type(self)(self.n * self._inv0(self._coerce(other)))

    def __rtruediv__(self, other):
        return type(self)(self._coerce(other) * self._inv0(self.n))

    def __pow__(self, e):
        e = int(e)
        if e < 0:
            return type(self)(pow(self._inv0(self.n), -e, self.field_modulus))
        return type(self)(pow(self.n, e, self.field_modulus))

    def __neg__(self):
        return type(self)(-self.n)

    def __eq__(self, other):
        if isinstance(other, _PrimeField):
            return self.n == other.n
        if isinstance(other, int):
            return self.n == other % self.field_modulus
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __hash__(self):
        return hash(self.n)

    def __int__(self):
        return self.n

    __index__ = __int__

    def __repr__(self):
        return "%s(%d)" % (type(self).__name__, self.n)

    @classmethod
    def one(cls):
        return cls(1)

    @classmethod
    def zero(cls):
        return cls(0)


class FQ(_PrimeField):
    """BN254 base field element (py_ecc.fields.bn128_FQ)."""
    field_modulus = FIELD_MODULUS
    __slots__ = ()


class FR(_PrimeField):
    """BN254 scalar field element (`class FR(FQ): field_modulus = bn128.curve_order`,
    zkp/plonk/field.py:36-51, zkp/groth16/proving.py:20-21)."""
    field_modulus = CURVE_ORDER
    __slots__ = ()


class FQ2:
    """F_p[i]/(i^2+1) element; `.coeffs` = (c0, c1) as FQ (py_ecc bn128_FQ2)."""
    __slots__ = ("coeffs",)

    def __init__(self, coeffs):
        c0, c1 = coeffs
        self.coeffs = (FQ(c0), FQ(c1))

    def __eq__(self, other):
        if isinstance(other, FQ2):
            return self.coeffs == other.coeffs
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __hash__(self):
        return hash(self.coeffs)

    def __neg__(self):
        return FQ2((-self.coeffs[0], -self.coeffs[1]))

    def __repr__(self):
        return "FQ2((%d, %d))" % (self.coeffs[0].n, self.coeffs[1].n)


G1 = (FQ(1), FQ(2))
G2 = (
    FQ2((10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634)),
    FQ2((8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531)),
)
Z1 = None  # point at infinity (zkp/plonk/field.py:69)


# ------------------------------------------------------------------ point <-> limb arrays
def is_g2(pt):
    return pt is not None and isinstance(pt[0], FQ2)


def _is_placeholder(pt):
    """(FQ(0), FQ(0)) placeholders of sigma1_3 / sigma1_4 (zkp/groth16/setup.py:39,50)."""
    return pt is not None and not is_g2(pt) and int(pt[0]) == 0 and int(pt[1]) == 0


def g1_to_limbs(points):
    """list of (FQ, FQ) | None -> (n, 8) uint64; None and (0,0) placeholders -> zeros (= infinity)."""
    flat = []
    for pt in points:
        if pt is None:
            flat += (0, 0)
        else:
            flat += (int(pt[0]), int(pt[1]))
    return _lib.ints_to_limbs(flat).reshape(len(points), 8)


def g2_to_limbs(points):
    flat = []
    for pt in points:
        if pt is None:
            flat += (0, 0, 0, 0)
        else:
            (x, y) = pt
            flat += (int(x.coeffs[0]), int(x.coeffs[1]), int(y.coeffs[0]), int(y.coeffs[1]))
    return _lib.ints_to_limbs(flat).reshape(len(points), 16)


def limbs_to_g1(arr):
    v = _lib.limbs_to_ints(arr)
    out = []
    for i in range(0, len(v), 2):
        out.append(None if v[i] == 0 and v[i + 1] == 0 else (FQ(v[i]), FQ(v[i + 1])))
    return out


def limbs_to_g2(arr):
    v = _lib.limbs_to_ints(arr)
    out = []
    for i in range(0, len(v), 4):
        q = v[i:i + 4]
        out.append(None if not any(q) else (FQ2((q[0], q[1])), FQ2((q[2], q[3]))))
    return out


def scalars_to_limbs(scalars):
    """FR / int scalars -> canonical (n, 4) limbs (reduced mod r as ec_mul does, field.py:86-88)."""
    return _lib.ints_to_limbs([int(s) % CURVE_ORDER for s in scalars])


# ------------------------------------------------------------------ MSM entry points
def msm_g1(scalars, points):
    """sum_i scalars[i] * points[i] on G1 through the GPU Pippenger pipeline (zk_msm_g1)."""
    n = len(points)
    if len(scalars) != n:
        raise ValueError("msm_g1: %d scalars for %d points" % (len(scalars), n))
    if n == 0:
        return None
    S, P = scalars_to_limbs(scalars), g1_to_limbs(points)
    out = np.zeros(8, dtype=np.uint64)
    inf = _lib.ctypes.c_int(0)
    _lib.check(_lib.load().zk_msm_g1(_lib.ptr(S), _lib.ptr(P), n, _lib.ptr(out), _lib.ctypes.byref(inf)))
    return None if inf.value else limbs_to_g1(out)[0]


def msm_g2(scalars, points):
    n = len(points)
    if len(scalars) != n:
        raise ValueError("msm_g2: %d scalars for %d points" % (len(scalars), n))
    if n == 0:
        return None
    S, P = scalars_to_limbs(scalars), g2_to_limbs(points)
    out = np.zeros(16, dtype=np.uint64)
    inf = _lib.ctypes.c_int(0)
    _lib.check(_lib.load().zk_msm_g2(_lib.ptr(S), _lib.ptr(P), n, _lib.ptr(out), _lib.ctypes.byref(inf)))
    return None if inf.value else limbs_to_g2(out)[0]


def msm(scalars, points):
    """Dispatch on the point type, as the reference's aliases do."""
    g2 = any(is_g2(p) for p in points)
    return msm_g2(scalars, points) if g2 else msm_g1(scalars, points)


def fixed_base_mul(point, scalars):
    """[k * point for k in scalars] in one GPU batch (zk_fixed_base_g1/g2).

    Replaces the one-multiply-per-element setup loops: zkp/groth16/setup.py:18-23,56-60,65-69,
    zkp/plonk/srs.py:77-85."""
    n = len(scalars)
    if n == 0:
        return []
    if point is None:
        return [None] * n
    S = scalars_to_limbs(scalars)
    lib = _lib.load()
    if is_g2(point):
        B, out = g2_to_limbs([point]), np.zeros((n, 16), dtype=np.uint64)
        _lib.check(lib.zk_fixed_base_g2(_lib.ptr(B), _lib.ptr(S), n, _lib.ptr(out)))
        return limbs_to_g2(out)
    B, out = g1_to_limbs([point]), np.zeros((n, 8), dtype=np.uint64)
    _lib.check(lib.zk_fixed_base_g1(_lib.ptr(B), _lib.ptr(S), n, _lib.ptr(out)))
    return limbs_to_g1(out)


# ------------------------------------------------------------------ reference seam
def ec_mul(point, scalar):
    """scalar * point (zkp/plonk/field.py:72-88; aliases `mult`, zkp/groth16/proving.py:12)."""
    if point is None:
        return None
    return fixed_base_mul(point, [scalar])[0]


def ec_add(p1, p2):
    """p1 + p2 (zkp/plonk/field.py:91-103; alias `add`, zkp/groth16/proving.py:14)."""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    lib = _lib.load()
    if is_g2(p1):
        A, B, out = g2_to_limbs([p1]), g2_to_limbs([p2]), np.zeros((1, 16), dtype=np.uint64)
        _lib.check(lib.zk_group_op(_lib.GROUP_G2, 0, _lib.ptr(A), _lib.ptr(B), 1, _lib.ptr(out)))
        return limbs_to_g2(out)[0]
    A, B, out = g1_to_limbs([p1]), g1_to_limbs([p2]), np.zeros((1, 8), dtype=np.uint64)
    _lib.check(lib.zk_group_op(_lib.GROUP_G1, 0, _lib.ptr(A), _lib.ptr(B), 1, _lib.ptr(out)))
    return limbs_to_g1(out)[0]


def ec_neg(point):
    """-point (zkp/plonk/field.py:106-115): y -> -y, a host-side sign flip."""
    if point is None:
        return None
    return (point[0], -point[1])


class FQ12:
    """Target-group element: 12 coefficients of F_p[w]/(w^12 - 18 w^6 + 82) (py_ecc bn128_FQ12 order).
    Supports what the verifiers use: ==, * and ** (zkp/groth16/verifying.py:29-40)."""
    __slots__ = ("coeffs",)

    def __init__(self, coeffs):
        self.coeffs = tuple(FQ(c) for c in coeffs)
        if len(self.coeffs) != 12:
            raise ValueError("FQ12 needs 12 coefficients")

    @classmethod
    def one(cls):
        return cls([1] + [0] * 11)

    def __mul__(self, other):
        if isinstance(other, (int, FQ)):
            return FQ12([c * other for c in self.coeffs])
        a, b = [c.n for c in self.coeffs], [c.n for c in other.coeffs]
        t = [0] * 23
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    t[i + j] += x * y
        for k in range(22, 11, -1):  # w^12 = 18 w^6 - 82
            if t[k]:
                t[k - 6] += 18 * t[k]
                t[k - 12] -= 82 * t[k]
        return FQ12(t[:12])

    __rmul__ = __mul__

    def __pow__(self, e):
        e = int(e)
        r, base = FQ12.one(), self
        while e:
            if e & 1:
                r = r * base
            base = base * base
            e >>= 1
        return r

    def __eq__(self, other):
        if isinstance(other, FQ12):
            return self.coeffs == other.coeffs
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __hash__(self):
        return hash(self.coeffs)

    def __repr__(self):
        return "FQ12(%r)" % ([c.n for c in self.coeffs],)


def ec_pairing(g2_point, g1_point):
    """e(g1_point, g2_point) -> FQ12; argument order (G2, G1) as py_ecc.bn128.pairing and
    zkp/plonk/field.py:118-138.  Runs on the host (zk_pairing): a verifier needs 2-4 pairings."""
    P = g1_to_limbs([g1_point])
    Q = g2_to_limbs([g2_point])
    out = np.zeros(48, dtype=np.uint64)
    _check_pairing_inputs(_lib.load().zk_pairing(_lib.ptr(P), _lib.ptr(Q), _lib.ptr(out)))
    return FQ12(_lib.limbs_to_ints(out))


def _check_pairing_inputs(rc):
    """py_ecc's pairing asserts is_on_curve for both arguments (bn128_pairing.py); libzkhip reports the same condition as
    ZK_ERR_INVALID, raised here as the reference's AssertionError so that callers see one behaviour."""
    if rc == _lib.ZK_ERR_INVALID:
        raise AssertionError(_lib.load().zk_last_error().decode("utf-8", "replace"))
    _lib.check(rc)


def pairing_check(pairs):
    """True iff prod_i e(P_i, Q_i) == 1 for pairs [(P_i in G1, Q_i in G2), ...] (one final exponentiation)."""
    if not pairs:
        return True
    P = g1_to_limbs([p for p, _ in pairs])
    Q = g2_to_limbs([q for _, q in pairs])
    ok = _lib.ctypes.c_int(0)
    _check_pairing_inputs(_lib.load().zk_pairing_check(_lib.ptr(P), _lib.ptr(Q), len(pairs), _lib.ctypes.byref(ok)))
    return bool(ok.value)


mult, add, neg, pairing = ec_mul, ec_add, ec_neg, ec_pairing  # zkp/groth16/proving.py:12-15 aliases


def get_root_of_unity(n):
    """Primitive n-th root of unity 5^((r-1)/n) (zkp/plonk/field.py:145-182)."""
    if n < 1 or (n & (n - 1)) != 0:
        raise ValueError("n must be a power of two: %d" % n)
    if n > (1 << 28):
        raise ValueError("n must be at most 2^28: %d" % n)
    if n == 1:
        return FR(1)
    return FR(5) ** ((CURVE_ORDER - 1) // n)


def get_roots_of_unity(n):
    """[1, w, w^2, ..., w^(n-1)] (zkp/plonk/field.py:185-209)."""
    omega = get_root_of_unity(n)
    roots, cur = [], FR(1)
    for _ in range(n):
        roots.append(cur)
        cur = cur * omega
    return roots
