"""CPU tests of the exact field/curve code the HIP kernels run (csrc/field.h, csrc/curve.h compiled
for the host in tests/hostmath) against the oracle: limb arithmetic, Montgomery conversions, XYZZ
formulas and every exceptional case (P+P, P+(-P), infinity operands)."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

import c_oracle as co
import py_ref as o

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", params=["libhostmath.so", "libhostmath_dbg.so"])
def hm(request):
    """Both builds: plain, and -DZK_FIELD_DEBUG (aborts the process if any operation is called
    outside the value/limb bounds that field.h's lazy reduction relies on)."""
    d = os.path.join(HERE, "hostmath")
    subprocess.check_call(["make", "-s", "-C", d])
    return ctypes.CDLL(os.path.join(d, request.param))


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fop(hm, which, op, a, b=0):
    A, B, O = co.to_limbs([a]), co.to_limbs([b]), np.zeros(4, dtype=np.uint64)
    hm.hm_field_op(which, op, P(A), P(B), P(O))
    return co.from_limbs(O)[0]


def test_field_ops(hm):
    rnd = random.Random(2)
    for which, m in ((0, o.P), (1, o.R)):
        vals = [0, 1, 2, m - 1, m - 2, (1 << 253), (1 << 253) - 1] + [rnd.randrange(m) for _ in range(300)]
        for i in range(len(vals) - 1):
            a, b = vals[i], vals[i + 1]
            assert fop(hm, which, 0, a, b) == (a + b) % m
            assert fop(hm, which, 1, a, b) == (a - b) % m
            assert fop(hm, which, 2, a, b) == a * b % m
            assert fop(hm, which, 4, a) == (-a) % m
            assert fop(hm, which, 5, a) == a * a % m
        for a in vals[1:12]:
            assert fop(hm, which, 3, a) == pow(a, -1, m)


def g1mul(hm, p, k):
    A, K, O = co.g1_to_arr([p]), co.to_limbs([k]), np.zeros(8, dtype=np.uint64)
    hm.hm_g1_mul(P(A), P(K), P(O))
    return co.g1_from_arr(O)[0]


def g2mul(hm, p, k):
    A, K, O = co.g2_to_arr([p]), co.to_limbs([k]), np.zeros(16, dtype=np.uint64)
    hm.hm_g2_mul(P(A), P(K), P(O))
    return co.g2_from_arr(O)[0]


def g1add(hm, mode, p, q, k1=1, k2=1):
    A, B, K1, K2, O = co.g1_to_arr([p]), co.g1_to_arr([q]), co.to_limbs([k1]), co.to_limbs([k2]), np.zeros(8, dtype=np.uint64)
    hm.hm_g1_add(mode, P(A), P(B), P(K1), P(K2), P(O))
    return co.g1_from_arr(O)[0]


def g2add(hm, mode, p, q, k1=1, k2=1):
    A, B, K1, K2, O = co.g2_to_arr([p]), co.g2_to_arr([q]), co.to_limbs([k1]), co.to_limbs([k2]), np.zeros(16, dtype=np.uint64)
    hm.hm_g2_add(mode, P(A), P(B), P(K1), P(K2), P(O))
    return co.g2_from_arr(O)[0]


def test_scalar_mul(hm):
    rnd = random.Random(3)
    for k in [0, 1, 2, 3, o.R - 1, o.R] + [rnd.randrange(o.R) for _ in range(6)]:
        assert g1mul(hm, o.G1, k) == co.g1_mul(o.G1, k)
        assert g2mul(hm, o.G2, k) == co.g2_mul(o.G2, k)
    assert g1mul(hm, None, 5) is None


def test_add_exceptional_cases(hm):
    A, B = o.g1_multiply(o.G1, 123), o.g1_multiply(o.G1, 456)
    for p, q in [(A, B), (A, A), (A, o.g1_neg(A)), (None, A), (A, None), (None, None)]:
        assert g1add(hm, 0, p, q) == o.g1_add(p, q)
    for k1, k2 in [(5, 7), (5, 5), (5, o.R - 5), (0, 5), (5, 0), (0, 0)]:
        assert g1add(hm, 1, A, A, k1, k2) == o.g1_multiply(A, (k1 + k2) % o.R)
    A2, B2 = o.g2_multiply(o.G2, 123), o.g2_multiply(o.G2, 456)
    for p, q in [(A2, B2), (A2, A2), (A2, o.g2_neg(A2)), (None, A2), (A2, None)]:
        assert g2add(hm, 0, p, q) == o.g2_add(p, q)
    for k1, k2 in [(5, 7), (5, 5), (5, o.R - 5), (0, 5), (5, 0)]:
        assert g2add(hm, 1, A2, A2, k1, k2) == o.g2_multiply(A2, (k1 + k2) % o.R)


def test_team_add(hm):
    """curve.h team4_add / team2_add (the bucket reduction's additions by four or two lanes), the lanes run as host threads that
    meet at a barrier in every exchange: the general case, acc == q (doubling in role 0), acc == -q, either or both operands infinity --
    on operands out of scalar multiplications (modes 2, 4) and out of the mixed addition, whose lazier coordinates are what the first
    reduction level is given (modes 3, 5); the _dbg build checks every product's contract."""
    rnd = random.Random(11)
    A, B = o.g1_multiply(o.G1, 123), o.g1_multiply(o.G1, 456)
    A2, B2 = o.g2_multiply(o.G2, 123), o.g2_multiply(o.G2, 456)
    pairs = [(5, 7), (5, 5), (5, o.R - 5), (0, 5), (5, 0), (0, 0), (1, 1), (1, o.R - 1)] + [(rnd.randrange(o.R), rnd.randrange(o.R)) for _ in range(4)]
    for k1, k2 in pairs:
        for m in (2, 4):
            assert g1add(hm, m, A, A, k1, k2) == o.g1_multiply(A, (k1 + k2) % o.R)
            assert g1add(hm, m, A, B, k1, k2) == o.g1_add(o.g1_multiply(A, k1), o.g1_multiply(B, k2))
            assert g2add(hm, m, A2, A2, k1, k2) == o.g2_multiply(A2, (k1 + k2) % o.R)
            assert g2add(hm, m, A2, B2, k1, k2) == o.g2_add(o.g2_multiply(A2, k1), o.g2_multiply(B2, k2))
            # lazy operands: (k1 A + B) + (k2 B + A), and with B = A: (k1 + 1) A + (k2 + 1) A -- equal / opposite operands again
            assert g1add(hm, m + 1, A, B, k1, k2) == o.g1_add(o.g1_multiply(A, (k1 + 1) % o.R), o.g1_multiply(B, (k2 + 1) % o.R))
            assert g1add(hm, m + 1, A, A, k1, k2) == o.g1_multiply(A, (k1 + k2 + 2) % o.R)
            assert g2add(hm, m + 1, A2, B2, k1, k2) == o.g2_add(o.g2_multiply(A2, (k1 + 1) % o.R), o.g2_multiply(B2, (k2 + 1) % o.R))
            assert g2add(hm, m + 1, A2, A2, k1, k2) == o.g2_multiply(A2, (k1 + k2 + 2) % o.R)


def test_small_mul(hm):
    A = o.g1_multiply(o.G1, 99)
    for k in (0, 1, 2, 255, 32768, 65535):
        Ai, O = co.g1_to_arr([A]), np.zeros(8, dtype=np.uint64)
        hm.hm_g1_small_mul(P(Ai), ctypes.c_uint32(k), P(O))
        assert co.g1_from_arr(O)[0] == o.g1_multiply(A, k)


def test_lazy_bounds_and_equality(hm):
    """Operands at the edge of fe_mul's contract: unnormalised sums, values up to 10m, 6x."""
    rnd = random.Random(4)
    for which, m in ((0, o.P), (1, o.R)):
        vals = [0, 1, m - 1, m - 2] + [rnd.randrange(m) for _ in range(200)]
        for i in range(len(vals) - 1):
            a, b = vals[i], vals[i + 1]
            assert fop(hm, which, 6, a, b) == (a + b) * (2 * b) % m
            assert fop(hm, which, 7, a, b) == pow(a - b, 3, m)
            assert fop(hm, which, 8, a) == 6 * a % m
            assert fop(hm, which, 9, a, b) == (1 if a == b else 0)
            assert fop(hm, which, 9, a, a) == 1


def test_device_host_conversions(hm):
    rnd = random.Random(5)
    for which, m in ((0, o.P), (1, o.R)):
        for a in [0, 1, m - 1] + [rnd.randrange(m) for _ in range(50)]:
            A, H, D = co.to_limbs([a]), np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
            hm.hm_host_roundtrip(which, P(A), P(H), P(D))
            assert co.from_limbs(H)[0] == a and co.from_limbs(D)[0] == a


def test_long_accumulation_chain_keeps_bounds(hm):
    """300 mixed additions in one accumulator (the bucket loop), with negations, a repeated point
    (doubling branch) and a cancelling pair, then the host epilogue conversion."""
    rnd = random.Random(6)
    n = 300
    ks = [rnd.randrange(1, o.R) for _ in range(n)]
    ks[10] = ks[9]
    pts = co.g1_from_arr(co.g1_fixed_base_arr(o.G1, co.to_limbs(ks)))
    neg = np.array([rnd.randrange(2) for _ in range(n)], dtype=np.uint8)
    neg[9] = neg[10] = 0
    neg[21], ks[21] = 1 - neg[20], ks[20]
    pts[21] = pts[20]
    total = sum((-k if s else k) for k, s in zip(ks, neg)) % o.R
    O = np.zeros(8, dtype=np.uint64)
    hm.hm_g1_accumulate(P(co.g1_to_arr(pts)), ctypes.c_uint32(n), P(neg), P(O))
    assert co.g1_from_arr(O)[0] == co.g1_mul(o.G1, total)
    # everything cancels -> infinity
    pts2 = [pts[0], pts[0], pts[1], pts[1]]
    neg2 = np.array([0, 1, 1, 0], dtype=np.uint8)
    hm.hm_g1_accumulate(P(co.g1_to_arr(pts2)), ctypes.c_uint32(4), P(neg2), P(O))
    assert not O.any()
    # negated entries on the two rare paths that keep q.y: first addition into an empty accumulator, doubling
    for sel, negs, k in (([0], [1], -ks[0]), ([0, 0], [1, 1], -2 * ks[0]), ([1, 0, 0], [0, 1, 1], ks[1] - 2 * ks[0])):
        hm.hm_g1_accumulate(P(co.g1_to_arr([pts[i] for i in sel])), ctypes.c_uint32(len(sel)), P(np.array(negs, dtype=np.uint8)), P(O))
        assert co.g1_from_arr(O)[0] == co.g1_mul(o.G1, k % o.R)


def test_long_g2_accumulation_chain_keeps_bounds(hm):
    """200 mixed additions in one G2 accumulator (its mixed addition keeps X below 4p and lets P, R and Q - X3 run above 2p: curve.h),
    with negations, a repeated point (doubling branch) and a cancelling pair; then the accumulator as an operand of a full addition
    and of a doubling, as the bucket reduction uses it.  The ZK_FIELD_DEBUG build aborts on any bound the formulas overstep."""
    rnd = random.Random(8)
    n = 200
    ks = [rnd.randrange(1, o.R) for _ in range(n)]
    ks[10] = ks[9]
    ks[21] = ks[20]
    pts = [co.g2_mul(o.G2, k) for k in ks]
    neg = np.array([rnd.randrange(2) for _ in range(n)], dtype=np.uint8)
    neg[9] = neg[10] = 0
    neg[21] = 1 - neg[20]
    total = sum((-k if s else k) for k, s in zip(ks, neg)) % o.R
    O = np.zeros(48, dtype=np.uint64)
    hm.hm_g2_accumulate(P(co.g2_to_arr(pts)), ctypes.c_uint32(n), P(neg), P(O))
    got = co.g2_from_arr(O)
    assert got[0] == co.g2_mul(o.G2, total)
    assert got[1] == got[2] == co.g2_mul(o.G2, 2 * total % o.R)
    # negated entries on the paths that keep q.y: first addition into an empty accumulator, doubling; everything cancelling
    for sel, negs, k in (([0], [1], -ks[0]), ([0, 0], [1, 1], -2 * ks[0]), ([1, 0, 0], [0, 1, 1], ks[1] - 2 * ks[0]), ([0, 0, 1, 1], [0, 1, 1, 0], 0)):
        hm.hm_g2_accumulate(P(co.g2_to_arr([pts[i] for i in sel])), ctypes.c_uint32(len(sel)), P(np.array(negs, dtype=np.uint8)), P(O))
        assert co.g2_from_arr(O)[0] == co.g2_mul(o.G2, k % o.R)


def test_sanitizer_build_is_clean():
    """the CPU path under sanitizers.  tests/hostmath/sanitize_check links the device math compiled for
    the host, the product's host code (csrc/pairing.hip) and the C oracle with -fsanitize=address,undefined
    (-fno-sanitize-recover=all: any report aborts), runs a fixed workload through all three and cross-checks them."""
    d = os.path.join(HERE, "hostmath")
    subprocess.check_call(["make", "-s", "-C", d, "sanitize_check"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(d, "sanitize_check")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().endswith("sanitize ok")


def test_single_chain_reduced_add_and_sub(hm):
    """fe_add_r2 / fe_sub_r2 (field.h; the NTT butterflies) decide their conditional +-2m from the top limbs and take a slow
    path when those cannot tell (probability 2^-22 per operation: random transforms never get there).  Against the two-chain
    forms they replace, same representative limb for limb: uniform operands, sums and differences within a few units and
    within a few 2^232 of the 2m boundary, and every top-limb constellation around the ambiguous ones with all-zero,
    all-one and random lower limbs."""
    rng = np.random.default_rng(2929)
    mask = (1 << 29) - 1

    def limbs(v):
        return np.array([(v >> (29 * i)) & mask for i in range(8)] + [v >> 232], dtype=np.uint32)

    def value(l):
        return sum(int(x) << (29 * i) for i, x in enumerate(l))

    for which, m in ((0, o.P), (1, o.R)):
        two_m = 2 * m
        k8 = two_m >> 232
        rand = lambda: int.from_bytes(rng.bytes(40), "little") % two_m
        cases = {0: [], 1: []}
        for _ in range(3000):
            cases[0].append((rand(), rand()))
            cases[1].append((rand(), rand()))
        deltas = [d for e in (0, 1, 2, 3, 1 << 29, 1 << 231, 1 << 232, (1 << 232) + 1, (1 << 232) - 1, 1 << 233) for d in (e, -e)]
        for _ in range(300):
            a = rand()
            for d in deltas:
                b = two_m - a + d
                if 0 <= b < two_m:
                    cases[0].append((a, b))
                b = a + d
                if 0 <= b < two_m:
                    cases[1].append((a, b))
        lows = [0, (1 << 232) - 1, 1, (1 << 232) - 2]
        for _ in range(200):
            a8 = int(rng.integers(0, k8 + 1))
            for e in (-3, -2, -1, 0, 1, 2, 3):
                for la in lows + [int.from_bytes(rng.bytes(29), "little")]:
                    for lb in lows + [int.from_bytes(rng.bytes(29), "little")]:
                        b8 = k8 - a8 + e                                   # add: a_8 + b_8 - (2m)_8 = e
                        a, b = (a8 << 232) | la, (b8 << 232) | lb
                        if b8 >= 0 and a < two_m and b < two_m:
                            cases[0].append((a, b))
                        b8 = a8 + e                                        # sub: a_8 - b_8 = -e
                        b = (b8 << 232) | lb
                        if b8 >= 0 and a < two_m and b < two_m:
                            cases[1].append((a, b))
        fast, ref = np.zeros(9, dtype=np.uint32), np.zeros(9, dtype=np.uint32)
        for op in (0, 1):
            amb = 0
            for a, b in cases[op]:
                la, lb = limbs(a), limbs(b)
                hm.hm_r2_pair(which, op, P(la), P(lb), P(fast), P(ref))    # (the debug build also checks every bound contract)
                assert np.array_equal(fast, ref), (which, op, hex(a), hex(b))
                want = (a + b) if op == 0 else (a - b)
                want = want - two_m if want >= two_m else (want + two_m if want < 0 else want)
                assert value(fast) == want and all(int(x) <= mask for x in fast[:8])
                t = (int(la[8]) + int(lb[8]) - k8) if op == 0 else (int(la[8]) - int(lb[8]))
                amb += (t in (-1, 0)) if op == 0 else (t == 0)
            assert amb > 1000                                              # the slow path was exercised, not just present
