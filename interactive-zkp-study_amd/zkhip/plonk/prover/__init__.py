"""PLONK prover on the GPU backend (mirrors zkp/plonk/prover/__init__.py:45-211; the rounds live in round1..round5.py, each with the reference's
`execute(state)` entry point, so `from zkhip.plonk.prover import round3; round3.execute(state)` works as in plonk_routes.py:615-698).

Same five rounds, transcript labels, blinding degrees and proof fields as the reference.  The
polynomial side uses the backend instead of coefficient algebra:
  * interpolation of the wire / accumulator columns: inverse NTT (GPU);
  * round 3: the quotient t = C / Z_H is computed pointwise on the coset 5*H' of a domain H' with
    |H'| >= deg t + 1 (coset NTTs of the operand polynomials, one coset inverse NTT of the result)
    instead of O(n^2) polynomial products and long division (round3.py:114-147);
    z(omega x) needs no extra transform (it is the same evaluation vector rotated by |H'|/n);
  * every commitment: one G1 MSM (GPU).
`blinding` lets tests inject the 9 blinding scalars the reference draws with secrets.randbelow
(round1.py:106, round2.py:77); by default they are random, so two proofs of the same witness differ.
"""
import secrets

from ...field import FR, CURVE_ORDER as R
from ..transcript import Transcript
from . import round1, round2, round3, round4, round5
from .common import COSET_K, linearisation_scalars  # noqa: F401  (re-exported: verifier.py, prover_device.py)


class Proof:
    FIELDS = ("a_comm", "b_comm", "c_comm", "z_comm", "t_lo_comm", "t_mid_comm", "t_hi_comm", "a_eval", "b_eval", "c_eval",
              "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval", "r_eval", "W_zeta_comm", "W_zeta_omega_comm")

    def __init__(self):
        for f in self.FIELDS:
            setattr(self, f, None)


class ProverState:
    def __init__(self, a_vals, b_vals, c_vals, public_inputs, preprocessed, srs, blinding=None):
        self.a_vals, self.b_vals, self.c_vals = ([FR(v) for v in col] for col in (a_vals, b_vals, c_vals))
        self.public_inputs = public_inputs
        self.preprocessed = preprocessed
        self.srs = srs
        self.transcript = Transcript()
        self.n, self.omega, self.domain = preprocessed.n, preprocessed.omega, preprocessed.domain
        self.a_poly = self.b_poly = self.c_poly = self.z_poly = None
        self.t_lo_poly = self.t_mid_poly = self.t_hi_poly = None
        self.beta = self.gamma = self.alpha = self.zeta = self.v = None
        self.pi_poly = None
        self.proof = Proof()
        self._blinding = list(blinding) if blinding is not None else None

    def _blind(self, count):
        if self._blinding is not None:
            out, self._blinding = self._blinding[:count], self._blinding[count:]
            if len(out) != count:
                raise ValueError("not enough blinding scalars supplied")
            return [FR(v) for v in out]
        return [FR(secrets.randbelow(R)) for _ in range(count)]

    def build_proof(self):
        return self.proof


def prove(circuit, a_vals, b_vals, c_vals, public_inputs, preprocessed, srs, blinding=None):
    """prove(circuit, a, b, c, public_inputs, preprocessed, srs) -> Proof  (prover/__init__.py:158-211).
    `public_inputs` is carried but unused, as in the reference (PI(x) = 0)."""
    state = ProverState(a_vals, b_vals, c_vals, public_inputs, preprocessed, srs, blinding)
    for rnd in (round1, round2, round3, round4, round5):
        rnd.execute(state)
    return state.build_proof()
