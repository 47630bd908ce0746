"""Synthetic MSM / NTT workloads with closed-form answers (bench.py, tools/, the -m gpu tests).

BASELINE.json's configs are "2^k random scalars / points"; a result on 2^26 points cannot be checked by an
oracle that walks the points, so the bases are generated as known multiples of the generator,
P_i = k_i * G1, and the MSM is checked against (sum_i s_i k_i mod r) * G1 -- one scalar multiplication.
Two families:
  * random k_i (host, numpy)                       -- sizes up to ~2^22;
  * arithmetic k_i = k0 + i*d (k0 < 2^63, d < 2^32) -- any size: the dot product needs no per-element
    big integers (16-bit pieces of the scalars, block-wise 64-bit sums; on the host or on the device).
Nothing here is on the product path: it only prepares inputs and expected values.
"""
import numpy as np

from . import _lib

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_R_LIMBS = [(R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
G1_GEN_LIMBS = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
ARITH_K0 = 0x1234567890ABCDEF >> 1
ARITH_D = 0x9E3779B1


def random_scalars(rng, n):
    """Uniform in [0, r): 254-bit rejection sampling -> (n, 4) uint64 limbs."""
    out = np.empty((n, 4), dtype=np.uint64)
    filled = 0
    while filled < n:
        m = int((n - filled) * 1.4) + 16
        cand = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64)
        cand[:, 3] &= np.uint64((1 << 62) - 1)
        lt = np.zeros(m, dtype=bool)
        eq = np.ones(m, dtype=bool)
        for i in (3, 2, 1, 0):
            lt |= eq & (cand[:, i] < np.uint64(_R_LIMBS[i]))
            eq &= cand[:, i] == np.uint64(_R_LIMBS[i])
        good = cand[lt]
        take = min(len(good), n - filled)
        out[filled:filled + take] = good[:take]
        filled += take
    return out


def random_scalars_device(n, device, seed):
    """(n, 4) int64 device tensor of scalars uniform below r, drawn on the device (no 2 GB host array at 2^26):
    62-bit top limbs are rejected unless strictly below r's top limb (the excluded band top == r_top has
    relative weight 2^-62), the lower limbs are uniform 64-bit words."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    out = torch.empty((n, 4), dtype=torch.int64, device=device)
    filled = 0
    r_top = _R_LIMBS[3]
    lo, hi = -(1 << 63), (1 << 63) - 1
    while filled < n:
        m = min(int((n - filled) * 1.4) + 1024, 1 << 24)
        cand = torch.randint(lo, hi, (m, 4), dtype=torch.int64, device=device, generator=gen)
        cand[:, 3] &= (1 << 62) - 1
        good = cand[cand[:, 3] < r_top]
        take = min(good.shape[0], n - filled)
        out[filled:filled + take] = good[:take]
        filled += take
    return out


def limbs_dot_mod_r(a, b):
    """sum_i a_i * b_i mod r on Python ints (closed-form MSM check for random k_i)."""
    ai, bi = _lib.limbs_to_ints(a), _lib.limbs_to_ints(b)
    acc = 0
    for x, y in zip(ai, bi):
        acc += x * y
    return acc % R_MOD


def fixed_base_points(lib, ks):
    """P_i = k_i * G1 for (n, 4) scalar limbs (zk_fixed_base_g1; HOST arrays)."""
    n = ks.shape[0]
    pts = np.zeros((n, 8), dtype=np.uint64)
    _lib.check(lib.zk_fixed_base_g1(_lib.ptr(G1_GEN_LIMBS), _lib.ptr(np.ascontiguousarray(ks)), n, _lib.ptr(pts)))
    return pts


def arithmetic_points(lib, n, k0=ARITH_K0, d=ARITH_D, first=0):
    """P_i = (k0 + (first + i) * d) * G1 for i < n, k0 < 2^63, d < 2^32, first + n <= 2^30 (no wrap, no reduction mod r)."""
    i = np.arange(first, first + n, dtype=np.uint64)
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = np.uint64(k0) + i * np.uint64(d)
    return fixed_base_points(lib, ks)


def _dot_from_piece_sums(cs, cis, k0, d):
    sum_s = sum(int(cs[j]) << (16 * j) for j in range(16))
    sum_is = sum(int(cis[j]) << (16 * j) for j in range(16))
    return k0 * sum_s + d * sum_is


def arithmetic_dot(scalars, k0=ARITH_K0, d=ARITH_D, first=0):
    """sum_i s_i * (k0 + (first + i) * d) mod r for HOST limbs (n, 4): 16-bit pieces of s, block-wise uint64 sums."""
    n = scalars.shape[0]
    pieces = np.ascontiguousarray(scalars).view(np.uint16).reshape(n, 16)
    total = 0
    blk = 1 << 16                                                     # 16-bit piece * index < 2^46, 2^16 of them < 2^62
    for lo in range(0, n, blk):
        pc = pieces[lo:lo + blk].astype(np.uint64)
        idx = np.arange(first + lo, first + lo + pc.shape[0], dtype=np.uint64)
        cs = pc.sum(axis=0, dtype=np.uint64)
        cis = (pc * idx[:, None]).sum(axis=0, dtype=np.uint64)
        total += _dot_from_piece_sums(cs, cis, k0, d)
    return total % R_MOD


def arithmetic_dot_device(d_scalars, k0=ARITH_K0, d=ARITH_D, first=0):
    """The same sum for an (n, 4) int64 DEVICE tensor: piece sums per 2^16-element block in int64 on the device
    (16-bit piece * index < 2^46, 2^16 of them < 2^62), the big-integer tail on the host."""
    import torch
    n = d_scalars.shape[0]
    total = 0
    blk = 1 << 16
    step = 1 << 22                                                    # elements per device pass (bounds the temporaries)
    shifts = torch.tensor([0, 16, 32, 48], dtype=torch.int64, device=d_scalars.device)
    for lo in range(0, n, step):
        s = d_scalars[lo:lo + step]
        m = s.shape[0]
        pc = ((s.unsqueeze(2) >> shifts) & 0xFFFF).reshape(m, 16)     # little-endian 16-bit pieces (arithmetic shift, then mask)
        idx = torch.arange(first + lo, first + lo + m, dtype=torch.int64, device=s.device)
        pad = (-m) % blk
        if pad:
            pc = torch.cat([pc, torch.zeros((pad, 16), dtype=torch.int64, device=s.device)])
            idx = torch.cat([idx, torch.zeros(pad, dtype=torch.int64, device=s.device)])
        cs = pc.view(-1, blk, 16).sum(dim=1).cpu().numpy()
        cis = (pc * idx.unsqueeze(1)).view(-1, blk, 16).sum(dim=1).cpu().numpy()
        for b in range(cs.shape[0]):
            total += _dot_from_piece_sums(cs[b], cis[b], k0, d)
    return total % R_MOD
