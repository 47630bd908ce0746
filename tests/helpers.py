"""Shared helpers for the test-suite (seeded inputs, limb conversion, oracle access)."""
import numpy as np

import c_oracle as co
import py_ref as pr


def rand_fr_limbs(rng, n):
    return co.to_limbs([int.from_bytes(rng.bytes(32), "little") % pr.R for _ in range(n)])


def rand_g1_limbs(rng, n):
    """n random G1 points k_i*G1 from the C oracle -> ((n,8) limbs, (n,4) k limbs).  Up to 4096 points every one is an
    oracle scalar multiplication; beyond, sums of two members of a 1024-point pool (oracle additions, ~100x cheaper)."""
    if n <= 4096:
        ks = rand_fr_limbs(rng, n)
        return co.g1_fixed_base_arr(pr.G1, ks), ks
    pool_k = [int.from_bytes(rng.bytes(32), "little") % pr.R for _ in range(1024)]
    pool = co.g1_fixed_base_arr(pr.G1, co.to_limbs(pool_k))
    ia, ib = rng.integers(0, 1024, size=n), rng.integers(0, 1024, size=n)
    out = np.zeros((n, 8), dtype=np.uint64)
    lib = co.lib()
    tmp = np.zeros(8, dtype=np.uint64)
    for t in range(n):
        lib.orc_g1_add(co._p(pool[ia[t]]), co._p(pool[ib[t]]), co._p(tmp))
        out[t] = tmp
    return out, co.to_limbs([(pool_k[a] + pool_k[b]) % pr.R for a, b in zip(ia, ib)])


def rand_g2_limbs(rng, n):
    ks = rand_fr_limbs(rng, n)
    out = np.zeros((n, 16), dtype=np.uint64)
    g2 = co.g2_to_arr([pr.G2])
    for i in range(n):
        o = np.zeros(16, dtype=np.uint64)
        k = ks[i:i + 1].copy()
        co.lib().orc_g2_mul(co._p(g2), co._p(k), co._p(o))
        out[i] = o
    return out, ks


def limb_row(v):
    return co.to_limbs([v])[0]


# P_i = (k0 + i*d) * G1 and sum_i s_i * (k0 + i*d) mod r for sizes where per-point oracle work is out of reach: one
# implementation, shared with bench.py and tools/ (zkhip/synthetic.py); its host and device forms are checked against
# Python big integers in tests/test_bench_launcher.py.
from zkhip.synthetic import arithmetic_dot, arithmetic_points as arithmetic_g1_points  # noqa: E402,F401


# Oracle-side expectations for the at-scale Groth16 prover (BASELINE.json configs[3]): oracle/scale_ref.py
from scale_ref import lagrange_at, r1cs_closed_form, r1cs_crs_scalars  # noqa: E402,F401
