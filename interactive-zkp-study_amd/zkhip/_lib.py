"""ctypes binding of libzkhip.so (the C ABI in include/zkhip.h) plus int <-> limb marshalling.

The product path has no CPU fallback: if the shared library is missing, or no HIP device is
visible, every compute call raises (ZkhipError / RuntimeError) instead of silently computing
on the host.
"""
import ctypes
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The in-tree build.  The A/B tools (tools/ab_ntt.py, tools/ab_msm.py) point LIB_PATH at another build of the same ABI before the
# first load(); no environment variable redirects the product path.
LIB_PATH = os.path.join(_PKG_DIR, "libzkhip.so")

ZK_OK = 0
ZK_ERR_INVALID = -1
ZK_ERR_HIP = -2
ZK_ERR_NO_DEVICE = -3
ZK_ERR_NOMEM = -4
GROUP_G1 = 1
GROUP_G2 = 2
NTT_PLAIN, NTT_BLOCKED_TW, NTT_TRANSPOSED = 0, 1, 2   # buffer layouts of zk_ntt_dev_io


class ZkhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libzkhip error %d: %s" % (code, msg))
        self.code = code


_lib = None

_VP = ctypes.c_void_p
_SZ = ctypes.c_size_t
_PROTOS = {
    "zk_last_error": (ctypes.c_char_p, []),
    "zk_version": (ctypes.c_int, []),
    "zk_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "zk_set_device": (ctypes.c_int, [ctypes.c_int]),
    "zk_msm_g1": (ctypes.c_int, [_VP, _VP, _SZ, _VP, ctypes.POINTER(ctypes.c_int)]),
    "zk_msm_g2": (ctypes.c_int, [_VP, _VP, _SZ, _VP, ctypes.POINTER(ctypes.c_int)]),
    "zk_msm_plan_create": (ctypes.c_int, [ctypes.c_int, _SZ, ctypes.POINTER(_VP)]),
    "zk_msm_plan_create_ex": (ctypes.c_int, [ctypes.c_int, _SZ, ctypes.c_int, ctypes.POINTER(_VP)]),
    "zk_msm_plan_destroy": (ctypes.c_int, [_VP]),
    "zk_msm_plan_window_bits": (ctypes.c_int, [_VP, _SZ]),
    "zk_msm_plan_max_in_flight": (ctypes.c_int, [_VP]),
    "zk_msm_plan_bind_points": (ctypes.c_int, [_VP, _VP, _SZ, _VP]),
    "zk_msm_submit_bound": (ctypes.c_int, [_VP, _VP, _SZ, _SZ, _VP, ctypes.POINTER(ctypes.c_int)]),
    "zk_msm_plan_profile": (ctypes.c_int, [_VP, ctypes.c_int]),
    "zk_msm_plan_stage_ms": (ctypes.c_int, [_VP, ctypes.POINTER(ctypes.c_float)]),
    "zk_msm_dev": (ctypes.c_int, [_VP, _VP, _VP, _SZ, _VP, ctypes.POINTER(ctypes.c_int), _VP]),
    "zk_msm_dev_partial": (ctypes.c_int, [_VP, _VP, _VP, _SZ, _VP, _VP]),
    "zk_msm_submit": (ctypes.c_int, [_VP, _VP, _VP, _SZ, _VP, ctypes.POINTER(ctypes.c_int)]),
    "zk_msm_collect": (ctypes.c_int, [_VP, ctypes.c_int, _VP, ctypes.POINTER(ctypes.c_int)]),
    "zk_msm_collect_partial": (ctypes.c_int, [_VP, ctypes.c_int, _VP]),
    "zk_msm_fold_partials": (ctypes.c_int, [ctypes.c_int, _VP, _SZ, _VP, ctypes.POINTER(ctypes.c_int)]),
    "zk_msm_partial_limbs": (ctypes.c_int, [ctypes.c_int]),
    "zk_ntt_fr": (ctypes.c_int, [_VP, ctypes.c_uint, ctypes.c_int, _VP]),
    "zk_cache_clear": (ctypes.c_int, []),
    "zk_cache_stats": (ctypes.c_int, [_VP]),
    "zk_ntt_plan_create": (ctypes.c_int, [ctypes.c_uint, ctypes.POINTER(_VP)]),
    "zk_ntt_plan_destroy": (ctypes.c_int, [_VP]),
    "zk_ntt_dev": (ctypes.c_int, [_VP, _VP, ctypes.c_int, _VP, _VP]),
    "zk_ntt_dev_padded": (ctypes.c_int, [_VP, _VP, _VP, _SZ, ctypes.c_int, _VP, _VP]),
    "zk_ntt_dev_multi": (ctypes.c_int, [_VP, ctypes.c_uint, _VP, _VP, _SZ, ctypes.c_int, _VP, _VP]),
    "zk_ntt_dev_batch": (ctypes.c_int, [_VP, _VP, ctypes.c_uint, ctypes.c_int, _VP]),
    "zk_ntt_dev_io": (ctypes.c_int, [_VP, _VP, _VP, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint64, _VP, ctypes.c_int, _VP]),
    "zk_ntt_twiddle_dev": (ctypes.c_int, [_VP, _VP, ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, _VP]),
    "zk_fr_quotient_dev": (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "zk_fr_spmv_dev": (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "zk_plonk_quotient_dev": (ctypes.c_int, [_VP, ctypes.POINTER(_VP), _VP, ctypes.c_uint, _VP, _VP, _VP, _SZ, _VP]),
    "zk_plonk_perm_factors_dev": (ctypes.c_int, [_VP, _VP, ctypes.POINTER(_VP), _VP, _VP, _SZ, _VP]),
    "zk_frvec_create": (ctypes.c_int, [ctypes.POINTER(_VP)]),
    "zk_frvec_destroy": (ctypes.c_int, [_VP]),
    "zk_fr_lincomb_dev": (ctypes.c_int, [_VP, ctypes.POINTER(_VP), _VP, ctypes.c_uint, _VP, _SZ, _VP]),
    "zk_fr_mul_dev": (ctypes.c_int, [_VP, _VP, _VP, _SZ, _VP]),
    "zk_fr_scale_powers_dev": (ctypes.c_int, [_VP, _VP, _SZ, _VP, _VP]),
    "zk_fr_scan_dev": (ctypes.c_int, [_VP, _VP, _SZ, ctypes.c_int, ctypes.c_int, _VP]),
    "zk_fr_eval_dev": (ctypes.c_int, [_VP, _VP, _VP, ctypes.c_uint, _VP, _VP, _VP]),
    "zk_fixed_base_g1": (ctypes.c_int, [_VP, _VP, _SZ, _VP]),
    "zk_fixed_base_g2": (ctypes.c_int, [_VP, _VP, _SZ, _VP]),
    "zk_fixed_base_g1_dev": (ctypes.c_int, [_VP, _VP, _SZ, _VP, _VP]),
    "zk_fixed_base_g2_dev": (ctypes.c_int, [_VP, _VP, _SZ, _VP, _VP]),
    "zk_group_op": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _VP, _VP, _SZ, _VP]),
    "zk_measure_rate": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "zk_pairing": (ctypes.c_int, [_VP, _VP, _VP]),
    "zk_pairing_check": (ctypes.c_int, [_VP, _VP, _SZ, ctypes.POINTER(ctypes.c_int)]),
}
EXPORTS = tuple(_PROTOS)


def load():
    """Loads libzkhip.so once; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libzkhip.so not found at %s -- build it with `make -C interactive-zkp-study_amd/csrc` "
                "(there is no CPU fallback)" % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own libamdhip64; when torch shares the process (device
        # buffers, streams, torch.distributed) it has to be loaded FIRST so that both resolve to one HIP
        # runtime -- a second runtime initialised later reports "No HIP GPUs are available".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        # The host-buffer entry points keep plans per thread (zk_cache_*); their C++ thread_local destructors would otherwise return
        # device memory while the process is already tearing the HIP runtime down.  Drop the calling (main) thread's cache first.
        import atexit
        atexit.register(lambda: lib.zk_cache_clear())
    return _lib


def check(rc):
    if rc != ZK_OK:
        raise ZkhipError(rc, load().zk_last_error().decode("utf-8", "replace"))


def cache_stats():
    """{ntt_builds, ntt_hits, msm_builds, msm_hits} of the host-buffer plan cache (zk_cache_stats; calling thread)."""
    out = np.zeros(4, dtype=np.uint64)
    check(load().zk_cache_stats(ptr(out)))
    return dict(zip(("ntt_builds", "ntt_hits", "msm_builds", "msm_hits"), (int(v) for v in out)))


def device_count():
    n = ctypes.c_int(0)
    check(load().zk_device_count(ctypes.byref(n)))
    return n.value


# ------------------------------------------------------------------ marshalling
def ints_to_limbs(vals, count=None):
    """Iterable of ints in [0, 2^256) -> (n, 4) uint64 little-endian limb array."""
    buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()
    if count is not None and arr.shape[0] != count:
        raise ValueError("expected %d elements" % count)
    return arr


def limbs_to_ints(arr):
    raw = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def ptr(arr):
    return arr.ctypes.data_as(ctypes.c_void_p)
