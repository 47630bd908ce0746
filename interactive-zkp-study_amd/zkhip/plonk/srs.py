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
