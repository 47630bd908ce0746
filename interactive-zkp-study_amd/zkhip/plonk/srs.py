"""Structured reference string (mirrors zkp/plonk/srs.py:36-87): powers of tau in G1 as one
fixed-base GPU batch."""
import hashlib

from ..field import FR, G1, G2, CURVE_ORDER, fixed_base_mul


class SRS:
    def __init__(self, g1_powers, g2_powers, max_degree):
        self.g1_powers = g1_powers
        self.g2_powers = g2_powers
        self.max_degree = max_degree

    @classmethod
    def generate(cls, max_degree, seed=None):
        """tau = sha256(str(seed)) mod r (srs.py:68-70); g1_powers[i] = tau^i * G1 (srs.py:77-82);
        g2_powers = [G2, tau*G2] (srs.py:85)."""
        if seed is not None:
            h = hashlib.sha256(str(seed).encode()).digest()
            tau_int = int.from_bytes(h, "big") % CURVE_ORDER
        else:
            import secrets
            tau_int = secrets.randbelow(CURVE_ORDER - 1) + 1
        tau = FR(tau_int)
        powers, cur = [], FR(1)
        for _ in range(max_degree + 1):
            powers.append(cur.n)
            cur = cur * tau
        g1_powers = fixed_base_mul(G1, powers)
        g2_powers = [G2] + fixed_base_mul(G2, [tau.n])
        return cls(g1_powers, g2_powers, max_degree)


class DeviceSRS:
    """The same string for provers at scale: [tau^i]_1 as ONE device array of canonical affine points (max_degree + 1 rows of 8
    limbs), never materialised as Python objects.  The exponents tau^i come from zk_fr_scale_powers_dev, the points from the
    fixed-base batch on device buffers (zk_fixed_base_g1_dev) -- zkp/plonk/srs.py:68-85 with both loops on the GPU: 2^20 powers in
    about 10 ms where `SRS.generate` builds a million FR objects and point tuples.  `g2_powers` stays the two-element host list
    the verifier reads.  `DevicePlonk(selectors, sigmas, srs.d_g1)` takes the array as it is."""

    def __init__(self, d_g1, g2_powers, max_degree, tau=None):
        self.d_g1, self.g2_powers, self.max_degree, self.tau = d_g1, g2_powers, max_degree, tau

    @classmethod
    def generate(cls, max_degree, seed=None, tau=None, keep_tau=False):
        """seed: tau = sha256(str(seed)) mod r as in SRS.generate; tau: given directly (tests, benchmarks with a closed-form check);
        neither: drawn with `secrets`.  keep_tau stores the toxic value on the object (never do that outside tests)."""
        import numpy as np
        import torch
        from .. import _lib
        from ..device import FrVec
        from ..field import g1_to_limbs
        if tau is None:
            if seed is not None:
                tau = int.from_bytes(hashlib.sha256(str(seed).encode()).digest(), "big") % CURVE_ORDER
            else:
                import secrets
                tau = secrets.randbelow(CURVE_ORDER - 1) + 1
        tau %= CURVE_ORDER
        n = max_degree + 1
        st = torch.cuda.current_stream().cuda_stream
        one = torch.from_numpy(np.array([[1, 0, 0, 0]], dtype=np.uint64).view(np.int64)).cuda()
        powers = one.repeat(n, 1)
        fv = FrVec()
        fv.scale_powers(powers.data_ptr(), n, tau, st)
        d_g1 = torch.empty((n, 8), dtype=torch.int64, device="cuda")
        base = g1_to_limbs([G1])
        _lib.check(_lib.load().zk_fixed_base_g1_dev(_lib.ptr(base), powers.data_ptr(), n, d_g1.data_ptr(), st))
        fv.close()
        return cls(d_g1, [G2] + fixed_base_mul(G2, [tau]), max_degree, tau if keep_tau else None)

    def to_host(self):
        """-> SRS with Python point tuples (small degrees: cross-checks against SRS.generate)."""
        import numpy as np
        from ..field import limbs_to_g1
        return SRS(limbs_to_g1(self.d_g1.cpu().numpy().view(np.uint64)), self.g2_powers, self.max_degree)
