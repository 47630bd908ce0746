"""GPU parity tests of the F_r vector primitives (zk_fr_lincomb_dev, zk_fr_mul_dev, zk_fr_scale_powers_dev,
zk_fr_scan_dev) against Python integers -- bit-exact, sizes straddling the 2048-element scan chunks and their recursion."""
import numpy as np
import pytest

import py_ref as o
from helpers import rand_fr_limbs
from zkhip import _lib
from zkhip.device import FrVec

pytestmark = pytest.mark.gpu
R = o.R


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def _ints(t):
    return _lib.limbs_to_ints(t.cpu().numpy().view(np.uint64))


@pytest.mark.parametrize("n", [1, 5, 2048, 2049, 70001])
def test_lincomb_and_mul(n):
    rng = np.random.default_rng(n)
    vecs = [rand_fr_limbs(rng, n) for _ in range(8)]
    vecs[1][0] = 0
    vecs[2][n - 1] = _lib.ints_to_limbs([R - 1])[0]
    ints = [_lib.limbs_to_ints(v) for v in vecs]
    d = [_dev(v) for v in vecs]
    coef = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(8)]
    coef[0], coef[1] = 0, R - 1
    const = int.from_bytes(rng.bytes(32), "little") % R
    out = _dev(np.zeros((n, 4), dtype=np.uint64))
    for k in (0, 1, 3, 8):
        FrVec.lincomb(out.data_ptr(), [t.data_ptr() for t in d[:k]], coef[:k], n, constant=const)
        assert _ints(out) == [(const + sum(coef[j] * ints[j][i] for j in range(k))) % R for i in range(n)]
    FrVec.lincomb(d[3].data_ptr(), [d[3].data_ptr(), d[4].data_ptr()], [2, R - 1], n)           # in place, no constant
    assert _ints(d[3]) == [(2 * a - b) % R for a, b in zip(ints[3], ints[4])]
    FrVec.mul(out.data_ptr(), d[5].data_ptr(), d[6].data_ptr(), n)
    assert _ints(out) == [a * b % R for a, b in zip(ints[5], ints[6])]
    with pytest.raises(_lib.ZkhipError):
        FrVec.lincomb(out.data_ptr(), [t.data_ptr() for t in d] + [d[0].data_ptr()], coef + [1], n)   # more than 8 inputs


@pytest.mark.parametrize("n", [1, 2, 7, 2047, 2048, 2049, 6000, 2048 * 2048 + 5])
def test_scans_and_powers(n):
    rng = np.random.default_rng(100 + n)
    big = n > 100000
    X = rand_fr_limbs(rng, n) if not big else np.ascontiguousarray(np.tile(rand_fr_limbs(rng, 4099), (n // 4099 + 1, 1))[:n])
    xs = _lib.limbs_to_ints(X)
    fv = FrVec()
    for product in (False, True):
        for reverse in (False, True):
            d = _dev(X)
            fv.scan(d.data_ptr(), n, product=product, reverse=reverse)
            got = _ints(d)
            seq = xs[::-1] if reverse else xs
            acc = 1 if product else 0
            want = []
            for v in seq:
                acc = acc * v % R if product else (acc + v) % R
                want.append(acc)
            if reverse:
                want = want[::-1]
            if big:
                idx = [0, 1, 2047, 2048, 2049, n // 2, n - 2049, n - 2, n - 1]
                assert [got[i] for i in idx] == [want[i] for i in idx]
            else:
                assert got == want
    if not big:
        g = int.from_bytes(rng.bytes(32), "little") % R
        d = _dev(X)
        fv.scale_powers(d.data_ptr(), n, g)
        assert _ints(d) == [v * pow(g, i, R) % R for i, v in enumerate(xs)]
        # evaluate(z) = sum_i c_i z^i (polynomial.py:85-106) as scale + running sum
        fv.scan(d.data_ptr(), n, product=False)
        assert _ints(d)[-1] == sum(v * pow(g, i, R) for i, v in enumerate(xs)) % R
    fv.close()


def test_synthetic_division_and_grand_product():
    """The two composite uses the primitives are there for."""
    import torch
    rng = np.random.default_rng(9)
    n = 5000
    fv = FrVec()
    c = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n)]
    z = int.from_bytes(rng.bytes(32), "little") % R
    # (p(x) - p(z)) / (x - z): q[i] = z^-(i+1) * sum_{j>i} c[j] z^j
    d = _dev(_lib.ints_to_limbs(c))
    fv.scale_powers(d.data_ptr(), n, z)
    fv.scan(d.data_ptr(), n, product=False, reverse=True)
    q = d[1:].contiguous()
    zi = pow(z, -1, R)
    fv.scale_powers(q.data_ptr(), n - 1, zi)
    FrVec.lincomb(q.data_ptr(), [q.data_ptr()], [zi], n - 1)
    want_q, rem = o.div_polys(c, [(-z) % R, 1])
    assert _ints(q) == want_q and rem[0] == sum(v * pow(z, i, R) for i, v in enumerate(c)) % R
    # z_i = prod_{j<i} f_j / g_j = (prefix products of f) * (suffix products of g) / (product of all g)
    f = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n)]
    g = [int.from_bytes(rng.bytes(32), "little") % R or 1 for _ in range(n)]
    df, dg = _dev(_lib.ints_to_limbs(f)), _dev(_lib.ints_to_limbs(g))
    fv.scan(df.data_ptr(), n, product=True)
    fv.scan(dg.data_ptr(), n, product=True, reverse=True)
    g_total = _ints(dg[:1])[0]
    acc = torch.empty_like(df)
    FrVec.mul(acc[1:].data_ptr(), df[:-1].contiguous().data_ptr(), dg[1:].contiguous().data_ptr(), n - 1)   # F_excl[i] * Gs[i], i >= 1
    FrVec.lincomb(acc[1:].data_ptr(), [acc[1:].data_ptr()], [pow(g_total, -1, R) * 1 % R], n - 1)
    got = _ints(acc[1:])
    want, cur = [], 1
    for i in range(1, n):
        cur = cur * f[i - 1] % R * pow(g[i - 1], -1, R) % R
        want.append(cur)
    # Gs[i] / G_total = 1 / prod_{j<i} g_j
    assert got == want
    fv.close()


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 65535, 65536, 65537, 200003])
def test_eval_at_a_point(n):
    """zk_fr_eval_dev: p(z) = sum_i c_i z^i for up to eight polynomials of different lengths at one point, against Horner's rule
    on Python integers (Polynomial.evaluate, zkp/plonk/polynomial.py:85-106); lengths around the 256-element blocks and the
    65536-element stride, z = 0, 1, r - 1 and random points, an empty polynomial, more than eight refused."""
    rng = np.random.default_rng(900 + n)
    counts = [n, max(n - 1, 0), min(n, 7), 0, n, n // 2 + 1, n, 1]
    base = rand_fr_limbs(rng, 4099)
    polys = [np.ascontiguousarray(np.tile(base, (c // 4099 + 1, 1))[:max(c, 1)]) for c in counts]
    polys[4] = polys[4].copy()
    polys[4][0] = 0
    polys[4][-1] = _lib.ints_to_limbs([R - 1])[0]
    ints = [_lib.limbs_to_ints(p)[:c] for p, c in zip(polys, counts)]
    d = [_dev(p) for p in polys]
    out = _dev(np.zeros((8, 4), dtype=np.uint64))
    fv = FrVec()
    for z in (0, 1, R - 1, int.from_bytes(rng.bytes(32), "little") % R, 5):
        for k in (8, 3, 1):
            fv.eval([(t.data_ptr(), c) for t, c in zip(d[:k], counts[:k])], z, out.data_ptr())
            got = _ints(out)[:k]
            want = []
            for cs in ints[:k]:
                acc = 0
                for c in reversed(cs):
                    acc = (acc * z + c) % R
                want.append(acc)
            assert got == want, (z, k)
    with pytest.raises(_lib.ZkhipError):
        fv.eval([(t.data_ptr(), c) for t, c in zip(d, counts)] + [(d[0].data_ptr(), 1)], 5, out.data_ptr())
    with pytest.raises(_lib.ZkhipError):
        fv.eval([(d[0].data_ptr(), 1)], R, out.data_ptr())      # point not canonical
    fv.close()
