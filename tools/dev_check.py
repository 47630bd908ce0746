"""Developer smoke check on a GPU box: product path (libzkhip via ctypes) vs the C oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
import c_oracle as co, py_ref as pr
from zkhip import _lib
import ctypes

lib = _lib.load()
rng = np.random.default_rng(7)

def rand_scalars(n):
    vals = [int.from_bytes(rng.bytes(32), "little") % pr.R for _ in range(n)]
    return co.to_limbs(vals)

def rand_points(n):
    ks = rand_scalars(n)
    return co.g1_fixed_base_arr(pr.G1, ks)

which = sys.argv[1:] or ["msm", "ntt", "g2"]
ok = True
if "msm" in which:
    for n in [1, 2, 5, 100, 1000, 5000, 20000]:
        S, P = rand_scalars(n), rand_points(n)
        if n >= 5:
            S[1] = 0; S[2] = co.to_limbs([1])[0]; S[3] = co.to_limbs([pr.R - 1])[0]; P[4] = 0
        exp = co.g1_msm_arr(S, P)
        out = np.zeros(8, dtype=np.uint64); inf = ctypes.c_int(0)
        t = time.time()
        rc = lib.zk_msm_g1(_lib.ptr(S), _lib.ptr(P), n, _lib.ptr(out), ctypes.byref(inf))
        dt = time.time() - t
        good = rc == 0 and np.array_equal(out, exp)
        ok &= good
        print("msm_g1 n=%d rc=%d %s (%.1f ms) %s" % (n, rc, "OK" if good else "MISMATCH", dt * 1e3, lib.zk_last_error().decode() if rc else ""), flush=True)
if "ntt" in which:
    for L in list(range(0, 15)) + [16, 17, 18]:
        n = 1 << L
        X = rand_scalars(n)
        w = pr.get_root_of_unity(n)
        for inv in (0, 1):
            exp = co.ntt_arr(X, w, bool(inv))
            got = X.copy()
            rc = lib.zk_ntt_fr(_lib.ptr(got), L, inv, None)
            good = rc == 0 and np.array_equal(got, exp)
            ok &= good
            print("ntt L=%d inv=%d rc=%d %s %s" % (L, inv, rc, "OK" if good else "MISMATCH", lib.zk_last_error().decode() if rc else ""), flush=True)
if "g2" in which:
    for n in [1, 3, 50, 600]:
        S = rand_scalars(n)
        ks = rand_scalars(n)
        P = np.zeros((n, 16), dtype=np.uint64)
        g2 = co.g2_to_arr([pr.G2])
        for i in range(n):
            o = np.zeros(16, dtype=np.uint64)
            co.lib().orc_g2_mul(co._p(g2), co._p(ks[i:i+1].copy()), co._p(o)); P[i] = o
        exp = co.g2_msm_arr(S, P)
        out = np.zeros(16, dtype=np.uint64); inf = ctypes.c_int(0)
        rc = lib.zk_msm_g2(_lib.ptr(S), _lib.ptr(P), n, _lib.ptr(out), ctypes.byref(inf))
        good = rc == 0 and np.array_equal(out, exp)
        ok &= good
        print("msm_g2 n=%d rc=%d %s %s" % (n, rc, "OK" if good else "MISMATCH", lib.zk_last_error().decode() if rc else ""), flush=True)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
