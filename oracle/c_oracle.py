"""oracle/c_oracle.py -- TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/libbn254_oracle.so
(the plain-C restatement, oracle/bn254_oracle.c).  numpy arrays of uint64 limbs in, ints out."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbn254_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_ntt.restype = ctypes.c_int
    return _lib


def to_limbs(vals, words=4):
    """list of ints -> (n, words) uint64 little-endian limb array."""
    out = np.zeros((len(vals), words), dtype=np.uint64)
    for i, v in enumerate(vals):
        v = int(v)
        for j in range(words):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def from_limbs(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(arr[i, j]) << (64 * j) for j in range(4)) for i in range(arr.shape[0])]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def g1_to_arr(pts):
    """list of (x,y)|None -> (n,8) uint64; None -> zeros."""
    flat = []
    for pt in pts:
        flat += [0, 0] if pt is None else [pt[0], pt[1]]
    return to_limbs(flat).reshape(len(pts), 8)


def g1_from_arr(a):
    v = from_limbs(a)
    out = []
    for i in range(0, len(v), 2):
        out.append(None if v[i] == 0 and v[i + 1] == 0 else (v[i], v[i + 1]))
    return out


def g2_to_arr(pts):
    flat = []
    for pt in pts:
        flat += [0, 0, 0, 0] if pt is None else [pt[0][0], pt[0][1], pt[1][0], pt[1][1]]
    return to_limbs(flat).reshape(len(pts), 16)


def g2_from_arr(a):
    v = from_limbs(a)
    out = []
    for i in range(0, len(v), 4):
        q = v[i:i + 4]
        out.append(None if not any(q) else ((q[0], q[1]), (q[2], q[3])))
    return out


def field_op(which, op, a, b=0):
    A, B, O = to_limbs([a]), to_limbs([b]), np.zeros(4, dtype=np.uint64)
    lib().orc_field_op(which, op, _p(A), _p(B), _p(O))
    return from_limbs(O)[0]


def g1_mul(pt, k):
    P, K, O = g1_to_arr([pt]), to_limbs([k]), np.zeros(8, dtype=np.uint64)
    lib().orc_g1_mul(_p(P), _p(K), _p(O))
    return g1_from_arr(O)[0]


def g1_add(p, q):
    P, Q, O = g1_to_arr([p]), g1_to_arr([q]), np.zeros(8, dtype=np.uint64)
    lib().orc_g1_add(_p(P), _p(Q), _p(O))
    return g1_from_arr(O)[0]


def g2_mul(pt, k):
    P, K, O = g2_to_arr([pt]), to_limbs([k]), np.zeros(16, dtype=np.uint64)
    lib().orc_g2_mul(_p(P), _p(K), _p(O))
    return g2_from_arr(O)[0]


def g2_add(p, q):
    P, Q, O = g2_to_arr([p]), g2_to_arr([q]), np.zeros(16, dtype=np.uint64)
    lib().orc_g2_add(_p(P), _p(Q), _p(O))
    return g2_from_arr(O)[0]


def g1_msm_arr(scalars, points):
    """scalars (n,4) uint64, points (n,8) uint64 -> (8,) uint64 affine (zeros = infinity)."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    points = np.ascontiguousarray(points, dtype=np.uint64)
    O = np.zeros(8, dtype=np.uint64)
    lib().orc_g1_msm(_p(scalars), _p(points), ctypes.c_size_t(scalars.shape[0]), _p(O))
    return O


def g1_msm_bucket_arr(scalars, points, c=12):
    """Bucket-method MSM (orc_g1_msm_bucket), same result as g1_msm_arr; c = window bits."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    points = np.ascontiguousarray(points, dtype=np.uint64)
    O = np.zeros(8, dtype=np.uint64)
    rc = lib().orc_g1_msm_bucket(_p(scalars), _p(points), ctypes.c_size_t(scalars.shape[0]), ctypes.c_uint(c), _p(O))
    if rc:
        raise RuntimeError("orc_g1_msm_bucket failed: %d" % rc)
    return O


def g1_msm_bucket_mt_arr(scalars, points, c=13, threads=8):
    """orc_g1_msm_bucket_mt: the bucket method with its windows spread over `threads` threads."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    points = np.ascontiguousarray(points, dtype=np.uint64)
    O = np.zeros(8, dtype=np.uint64)
    rc = lib().orc_g1_msm_bucket_mt(_p(scalars), _p(points), ctypes.c_size_t(scalars.shape[0]), ctypes.c_uint(c), ctypes.c_uint(threads), _p(O))
    if rc:
        raise RuntimeError("orc_g1_msm_bucket_mt failed: %d" % rc)
    return O


def g2_msm_arr(scalars, points):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    points = np.ascontiguousarray(points, dtype=np.uint64)
    O = np.zeros(16, dtype=np.uint64)
    lib().orc_g2_msm(_p(scalars), _p(points), ctypes.c_size_t(scalars.shape[0]), _p(O))
    return O


def g2_msm_bucket_arr(scalars, points, c=12):
    """Bucket-method G2 MSM (orc_g2_msm_bucket), same result as g2_msm_arr; c = window bits."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    points = np.ascontiguousarray(points, dtype=np.uint64)
    O = np.zeros(16, dtype=np.uint64)
    rc = lib().orc_g2_msm_bucket(_p(scalars), _p(points), ctypes.c_size_t(scalars.shape[0]), ctypes.c_uint(c), _p(O))
    if rc:
        raise RuntimeError("orc_g2_msm_bucket failed: %d" % rc)
    return O


def g2_fixed_base_arr(pt, scalars):
    """out[i] = k_i * pt in G2; scalars (n,4) uint64 -> (n,16) uint64."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    P = g2_to_arr([pt])
    O = np.zeros((scalars.shape[0], 16), dtype=np.uint64)
    lib().orc_g2_fixed_base(_p(P), _p(scalars), ctypes.c_size_t(scalars.shape[0]), _p(O))
    return O


def g1_fixed_base_arr(pt, scalars):
    """out[i] = k_i * pt; scalars (n,4) uint64 -> (n,8) uint64."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    P = g1_to_arr([pt])
    O = np.zeros((scalars.shape[0], 8), dtype=np.uint64)
    lib().orc_g1_fixed_base(_p(P), _p(scalars), ctypes.c_size_t(scalars.shape[0]), _p(O))
    return O


def ntt_arr(data, omega, inverse=False):
    """data (n,4) uint64 canonical -> transformed copy."""
    d = np.array(data, dtype=np.uint64, copy=True).reshape(-1, 4)
    n = d.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    W = to_limbs([omega])
    rc = lib().orc_ntt(_p(d), ctypes.c_uint(log_n), _p(W), ctypes.c_int(1 if inverse else 0))
    assert rc == 0
    return d


def fr_horner_arr(coeffs, x):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    X, O = to_limbs([x]), np.zeros(4, dtype=np.uint64)
    lib().orc_fr_horner(_p(coeffs), ctypes.c_size_t(coeffs.shape[0]), _p(X), _p(O))
    return from_limbs(O)[0]


def fr_dot_arr(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    O = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_dot(_p(a), _p(b), ctypes.c_size_t(a.shape[0]), _p(O))
    return from_limbs(O)[0]


def fr_spmv_arr(row_ptr, col, vals, x):
    """y = M x over F_r, M in CSR form (row_ptr uint32[rows+1], col uint32[nnz], vals (nnz,4) uint64), x (n,4) uint64 -> (rows,4)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    vals = np.ascontiguousarray(vals, dtype=np.uint64)
    x = np.ascontiguousarray(x, dtype=np.uint64)
    rows = row_ptr.shape[0] - 1
    y = np.zeros((rows, 4), dtype=np.uint64)
    lib().orc_fr_spmv(_p(row_ptr), _p(col), _p(vals), _p(x), ctypes.c_size_t(rows), _p(y))
    return y

