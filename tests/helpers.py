"""Shared helpers for the test-suite (seeded inputs, limb conversion, oracle access)."""
import numpy as np

import c_oracle as co
import py_ref as pr


def rand_fr_limbs(rng, n):
    return co.to_limbs([int.from_bytes(rng.bytes(32), "little") % pr.R for _ in range(n)])


def rand_g1_limbs(rng, n):
    """n random G1 points k_i*G1 (C oracle fixed base) -> ((n,8) limbs, (n,4) k limbs)."""
    ks = rand_fr_limbs(rng, n)
    return co.g1_fixed_base_arr(pr.G1, ks), ks


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
