"""Device-resident entry points: thin handle classes over the plan API of include/zkhip.h.

Buffers are DEVICE pointers (e.g. torch tensor .data_ptr()); streams are raw hipStream_t
handles (e.g. torch.cuda.current_stream().cuda_stream).  Used by bench.py and the at-scale
prover; the list-based facade (zkhip.field / groth16 / plonk) goes through the host-buffer calls.
"""
import ctypes

import numpy as np

from . import _lib
from .field import limbs_to_g1, limbs_to_g2


class MsmPlan:
    """Workspace + pipeline for G1/G2 MSMs of up to max_n points (zk_msm_plan_*)."""

    def __init__(self, group, max_n, chunk_log=0):
        """chunk_log: MSMs of more than 2^chunk_log points run as chunks of that size (0 = the library's 2^22;
        zk_msm_plan_create_ex)."""
        self.group = group
        self.max_n = int(max_n)
        self._h = ctypes.c_void_p()
        if chunk_log:
            _lib.check(_lib.load().zk_msm_plan_create_ex(group, self.max_n, int(chunk_log), ctypes.byref(self._h)))
        else:
            _lib.check(_lib.load().zk_msm_plan_create(group, self.max_n, ctypes.byref(self._h)))
        self._limbs = 8 if group == _lib.GROUP_G1 else 16

    def close(self):
        if self._h:
            _lib.load().zk_msm_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def window_bits(self, n):
        return _lib.load().zk_msm_plan_window_bits(self._h, n)

    def set_profiling(self, enable):
        _lib.check(_lib.load().zk_msm_plan_profile(self._h, 1 if enable else 0))

    def bind(self, d_points, n, stream=0):
        """Expand n device points into the plan's table (zk_msm_plan_bind_points); afterwards pass d_points=None / 0 to
        run / submit to use the bound bases (13 n bucket additions instead of 16 n).  n = 0 unbinds."""
        _lib.check(_lib.load().zk_msm_plan_bind_points(self._h, d_points, n, stream))

    def max_in_flight(self):
        """Submissions that may be outstanding (submit without collect) on this plan."""
        return _lib.load().zk_msm_plan_max_in_flight(self._h)

    def stage_ms(self):
        """Device time of the last run: (prepare, sort, accumulate, reduce) in ms (HIP events on the
        pipeline's stream; needs set_profiling(True))."""
        out = (ctypes.c_float * 4)()
        _lib.check(_lib.load().zk_msm_plan_stage_ms(self._h, out))
        return tuple(float(v) for v in out)

    def run_limbs(self, d_scalars, d_points, n, stream=0):
        """-> (uint64[8|16] canonical affine limbs, is_inf)."""
        out = np.zeros(self._limbs, dtype=np.uint64)
        inf = ctypes.c_int(0)
        _lib.check(_lib.load().zk_msm_dev(self._h, d_scalars, d_points, n, _lib.ptr(out), ctypes.byref(inf), stream))
        return out, bool(inf.value)

    def run(self, d_scalars, d_points, n, stream=0):
        out, inf = self.run_limbs(d_scalars, d_points, n, stream)
        if inf:
            return None
        return (limbs_to_g1(out) if self.group == _lib.GROUP_G1 else limbs_to_g2(out))[0]

    # pipelined form: up to max_in_flight() submissions outstanding; consecutive MSMs overlap on the GPU and the host
    # fold of one hides behind the next
    def submit(self, d_scalars, d_points, n, stream=0):
        t = ctypes.c_int(-1)
        _lib.check(_lib.load().zk_msm_submit(self._h, d_scalars, d_points, n, stream, ctypes.byref(t)))
        return t.value

    def submit_bound(self, d_scalars, first, n, stream=0):
        """Pipelined MSM over the bound bases [first, first + n) (zk_msm_submit_bound)."""
        t = ctypes.c_int(-1)
        _lib.check(_lib.load().zk_msm_submit_bound(self._h, d_scalars, first, n, stream, ctypes.byref(t)))
        return t.value

    def collect_limbs(self, ticket):
        out = np.zeros(self._limbs, dtype=np.uint64)
        inf = ctypes.c_int(0)
        _lib.check(_lib.load().zk_msm_collect(self._h, ticket, _lib.ptr(out), ctypes.byref(inf)))
        return out, bool(inf.value)

    def collect_partial(self, ticket):
        out = np.zeros(2 * self._limbs, dtype=np.uint64)
        _lib.check(_lib.load().zk_msm_collect_partial(self._h, ticket, _lib.ptr(out)))
        return out

    def run_partial(self, d_scalars, d_points, n, stream=0):
        """-> uint64[16|32]: this device's partial sum in XYZZ Montgomery limbs (for folding)."""
        out = np.zeros(2 * self._limbs, dtype=np.uint64)
        _lib.check(_lib.load().zk_msm_dev_partial(self._h, d_scalars, d_points, n, _lib.ptr(out), stream))
        return out


_NO_MULTI = bool(__import__("os").environ.get("ZK_NTT_NO_MULTI"))


class NttPlan:
    """Twiddle tables + scratch for in-place device NTTs of size 2^log_n (zk_ntt_plan_*)."""

    def __init__(self, log_n):
        self.log_n = int(log_n)
        self._h = ctypes.c_void_p()
        _lib.check(_lib.load().zk_ntt_plan_create(self.log_n, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            _lib.load().zk_ntt_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, d_data, inverse=False, coset_shift=None, stream=0):
        k = None if coset_shift is None else _lib.ints_to_limbs([int(coset_shift)])
        _lib.check(_lib.load().zk_ntt_dev(self._h, d_data, 1 if inverse else 0, None if k is None else _lib.ptr(k), stream))


    def run_padded(self, d_in, d_out, in_len, inverse=False, coset_shift=None, stream=0):
        """The transform of an input that is zero from element in_len on, from d_in to d_out (may be the same buffer): only the
        first in_len elements of d_in are read (zk_ntt_dev_padded) -- no zero fill, no copy into a domain-sized buffer."""
        k = None if coset_shift is None else _lib.ints_to_limbs([int(coset_shift)])
        _lib.check(_lib.load().zk_ntt_dev_padded(self._h, d_in, d_out, int(in_len), 1 if inverse else 0, None if k is None else _lib.ptr(k), stream))

    def run_multi(self, pairs, in_len=None, inverse=False, coset_shift=None, stream=0):
        """Up to four independent transforms in one launch per pass (zk_ntt_dev_multi): pairs = [(d_in, d_out), ...] device pointers,
        d_out == d_in allowed; in_len (default n): elements read from every input, the rest counts as zero."""
        if _NO_MULTI:   # measurement hook (profiles/r05_experiments.md): the same transforms one after the other
            for d_in, d_out in pairs:
                self.run_padded(d_in, d_out, (1 << self.log_n) if in_len is None else in_len, inverse, coset_shift, stream)
            return
        k = len(pairs)
        ins = (ctypes.c_void_p * k)(*[p[0] for p in pairs])
        outs = (ctypes.c_void_p * k)(*[p[1] for p in pairs])
        shift = None if coset_shift is None else _lib.ints_to_limbs([int(coset_shift)])
        _lib.check(_lib.load().zk_ntt_dev_multi(self._h, k, ins, outs, (1 << self.log_n) if in_len is None else int(in_len), 1 if inverse else 0,
                                                None if shift is None else _lib.ptr(shift), stream))

    def run_batch(self, d_data, batch, inverse=False, stream=0):
        """`batch` independent transforms stored back to back in d_data (zk_ntt_dev_batch)."""
        _lib.check(_lib.load().zk_ntt_dev_batch(self._h, d_data, int(batch), 1 if inverse else 0, stream))

    def run_io(self, d_in, d_out, batch, inverse=False, in_layout=0, out_layout=0, log_block=0, row0=0, big=None, tw_inverse=False, stream=0):
        """`batch` transforms from d_in to d_out in the four-step layouts (_lib.NTT_PLAIN / NTT_BLOCKED_TW / NTT_TRANSPOSED;
        zk_ntt_dev_io): pack, transpose and the twiddle of the large transform `big` happen in the first loads / last stores."""
        _lib.check(_lib.load().zk_ntt_dev_io(self._h, d_in, d_out, int(batch), 1 if inverse else 0, int(in_layout), int(out_layout), int(log_block),
                                              int(row0), None if big is None else big._h, 1 if tw_inverse else 0, stream))

    def twiddle(self, d_data, log_cols, rows, row0, inverse=False, stream=0):
        """d_data[b * 2^log_cols + k] *= omega_n^(+-(row0 + b) * k) for b < rows, n = this plan's size (zk_ntt_twiddle_dev)."""
        _lib.check(_lib.load().zk_ntt_twiddle_dev(self._h, d_data, int(log_cols), int(rows), int(row0), 1 if inverse else 0, stream))


def fr_quotient(d_out, d_a, d_b, d_c, zinv, n, stream=0):
    """out[i] = (a[i]*b[i] - c[i]) * zinv on device buffers (zk_fr_quotient_dev)."""
    z = _lib.ints_to_limbs([int(zinv)])
    _lib.check(_lib.load().zk_fr_quotient_dev(d_out, d_a, d_b, d_c, _lib.ptr(z), n, stream))


def fr_spmv(d_row_ptr, d_col, d_vals, d_x, d_y, rows, stream=0):
    """y = M x over F_r, M in CSR form on device buffers (zk_fr_spmv_dev)."""
    _lib.check(_lib.load().zk_fr_spmv_dev(d_row_ptr, d_col, d_vals, d_x, d_y, rows, stream))


class FrVec:
    """F_r vector primitives on device buffers (zk_fr_lincomb_dev / _mul_dev / _scale_powers_dev / _scan_dev).
    Arguments are raw device pointers to canonical elements (e.g. torch.Tensor.data_ptr() of an (n, 4) int64 tensor)."""

    def __init__(self):
        self._h = ctypes.c_void_p()
        _lib.check(_lib.load().zk_frvec_create(ctypes.byref(self._h)))

    def close(self):
        if self._h:
            _lib.load().zk_frvec_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def lincomb(d_out, d_ins, coeffs, n, constant=None, stream=0):
        """out[i] = constant + sum_j coeffs[j] * ins[j][i]  (at most 8 inputs; out may alias an input)."""
        k = len(d_ins)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[ctypes.c_void_p(int(p)) for p in d_ins])
        cf = _lib.ints_to_limbs([int(c) for c in coeffs]) if k else np.zeros((1, 4), dtype=np.uint64)
        cst = None if constant is None else _lib.ints_to_limbs([int(constant)])
        _lib.check(_lib.load().zk_fr_lincomb_dev(d_out, ptrs, _lib.ptr(cf), k, None if cst is None else _lib.ptr(cst), n, stream))

    @staticmethod
    def mul(d_out, d_a, d_b, n, stream=0):
        _lib.check(_lib.load().zk_fr_mul_dev(d_out, d_a, d_b, n, stream))

    def scale_powers(self, d_data, n, base, stream=0):
        """data[i] *= base^i."""
        b = _lib.ints_to_limbs([int(base)])
        _lib.check(_lib.load().zk_fr_scale_powers_dev(self._h, d_data, n, _lib.ptr(b), stream))

    def eval(self, items, point, d_out, stream=0):
        """d_out[j] = sum_i coefs_j[i] * point^i for items = [(d_coefs, count), ...] (at most 8) at ONE point (zk_fr_eval_dev):
        one pass over the coefficients; the k values stay on the device (d_out: k canonical elements)."""
        k = len(items)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[ctypes.c_void_p(int(p)) for p, _ in items])
        cnt = (ctypes.c_size_t * max(k, 1))(*[int(c) for _, c in items])
        z = _lib.ints_to_limbs([int(point)])
        _lib.check(_lib.load().zk_fr_eval_dev(self._h, ptrs, cnt, k, _lib.ptr(z), d_out, stream))

    def scan(self, d_data, n, product=False, reverse=False, stream=0):
        """In-place inclusive scan: running sums (product=False) or products; reverse: from the last element down."""
        _lib.check(_lib.load().zk_fr_scan_dev(self._h, d_data, n, 1 if product else 0, 1 if reverse else 0, stream))


def plonk_perm_factors(d_num, d_den, d_ins, beta, gamma, n, stream=0):
    """The per-row numerators / denominators of PLONK's grand product in one pass (zk_plonk_perm_factors_dev); d_ins: the 7 device
    vectors a b c | s1 s2 s3 | x (x[i] = omega^i)."""
    ptrs = (ctypes.c_void_p * 7)(*[ctypes.c_void_p(int(p)) for p in d_ins])
    sc = _lib.ints_to_limbs([int(beta), int(gamma)])
    _lib.check(_lib.load().zk_plonk_perm_factors_dev(d_num, d_den, ptrs, _lib.ptr(sc[0:1]), _lib.ptr(sc[1:2]), n, stream))


def plonk_quotient(d_out, d_ins, zh_inv, alpha, beta, gamma, n, stream=0):
    """Fused PLONK round-3 quotient on the evaluation coset (zk_plonk_quotient_dev); d_ins: the 15 device vectors
    a b c z zw | q_L q_R q_O q_M q_C | s1 s2 s3 | x L1, zh_inv: the `period` values of 1 / Z_H."""
    ptrs = (ctypes.c_void_p * 15)(*[ctypes.c_void_p(int(p)) for p in d_ins])
    zi = _lib.ints_to_limbs([int(v) for v in zh_inv])
    sc = _lib.ints_to_limbs([int(alpha), int(beta), int(gamma)])
    _lib.check(_lib.load().zk_plonk_quotient_dev(d_out, ptrs, _lib.ptr(zi), len(zh_inv), _lib.ptr(sc[0:1]), _lib.ptr(sc[1:2]), _lib.ptr(sc[2:3]), n, stream))
