"""CPU tests of the drop-in boundary: libzkhip.so loads, exports every symbol include/zkhip.h
declares, fails loudly without a device (no CPU fallback), and its host-only epilogue
(zk_msm_fold_partials) is exact."""
import ctypes
import os
import re

import numpy as np
import pytest

import c_oracle as co
import py_ref as o
from zkhip import _lib
from zkhip.distributed import fold_partials, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MONT = 1 << 256


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), "libzkhip.so does not export %s" % name
    # the Python binding covers the same surface
    assert sorted(_lib.EXPORTS) == names
    assert _lib.load().zk_version() == 1


def _has_device():
    return _lib.device_count() > 0


def test_compute_entry_points_fail_loudly_without_a_device():
    if _has_device():
        pytest.skip("a HIP device is present")
    lib = _lib.load()
    s = np.zeros((1, 4), dtype=np.uint64)
    p = np.zeros((1, 8), dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    inf = ctypes.c_int(0)
    assert lib.zk_msm_g1(_lib.ptr(s), _lib.ptr(p), 1, _lib.ptr(out), ctypes.byref(inf)) == _lib.ZK_ERR_NO_DEVICE
    assert b"no HIP device" in lib.zk_last_error()
    d = np.zeros((4, 4), dtype=np.uint64)
    assert lib.zk_ntt_fr(_lib.ptr(d), 2, 0, None) == _lib.ZK_ERR_NO_DEVICE
    h = ctypes.c_void_p()
    assert lib.zk_msm_plan_create(_lib.GROUP_G1, 16, ctypes.byref(h)) == _lib.ZK_ERR_NO_DEVICE
    from zkhip.field import G1, ec_mul
    with pytest.raises(_lib.ZkhipError):
        ec_mul(G1, 5)


def test_argument_validation():
    lib = _lib.load()
    out = np.zeros(8, dtype=np.uint64)
    assert lib.zk_msm_g1(None, None, 3, _lib.ptr(out), None) == _lib.ZK_ERR_INVALID
    assert lib.zk_ntt_fr(None, 2, 0, None) == _lib.ZK_ERR_INVALID
    assert lib.zk_msm_partial_limbs(_lib.GROUP_G1) == 16 and lib.zk_msm_partial_limbs(_lib.GROUP_G2) == 32
    assert lib.zk_msm_partial_limbs(7) == _lib.ZK_ERR_INVALID


def xyzz_partial_g1(pt):
    """Affine oracle point -> XYZZ Montgomery limbs as zk_msm_dev_partial writes them."""
    if pt is None:
        return np.zeros(16, dtype=np.uint64)
    one = MONT % o.P
    return co.to_limbs([pt[0] * MONT % o.P, pt[1] * MONT % o.P, one, one]).reshape(16)


def xyzz_partial_g2(pt):
    if pt is None:
        return np.zeros(32, dtype=np.uint64)
    m = lambda v: v * MONT % o.P
    (x0, x1), (y0, y1) = pt
    return co.to_limbs([m(x0), m(x1), m(y0), m(y1), MONT % o.P, 0, MONT % o.P, 0]).reshape(32)


def test_fold_partials_host_epilogue():
    pts = [o.g1_multiply(o.G1, k) for k in (3, 5, 7, 11)]
    exp = None
    for p in pts:
        exp = o.g1_add(exp, p)
    parts = np.stack([xyzz_partial_g1(p) for p in pts])
    got = fold_partials(_lib.GROUP_G1, parts)
    assert (int(got[0]), int(got[1])) == exp
    # infinity partials, doubling, cancellation
    assert fold_partials(_lib.GROUP_G1, np.stack([xyzz_partial_g1(None), xyzz_partial_g1(None)])) is None
    got = fold_partials(_lib.GROUP_G1, np.stack([xyzz_partial_g1(pts[0]), xyzz_partial_g1(pts[0])]))
    assert (int(got[0]), int(got[1])) == o.g1_double(pts[0])
    assert fold_partials(_lib.GROUP_G1, np.stack([xyzz_partial_g1(pts[0]), xyzz_partial_g1(o.g1_neg(pts[0]))])) is None
    q = [o.g2_multiply(o.G2, k) for k in (2, 9)]
    got = fold_partials(_lib.GROUP_G2, np.stack([xyzz_partial_g2(p) for p in q]))
    exp2 = o.g2_add(q[0], q[1])
    assert tuple(int(c) for c in got[0].coeffs) == exp2[0] and tuple(int(c) for c in got[1].coeffs) == exp2[1]


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
