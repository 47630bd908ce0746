"""zkhip -- MI355X-native MSM / NTT backend behind the setup / prove / commit / fft call
signatures of tokamak-network/interactive-zkp-study (zkp.groth16.*, zkp.plonk.*).

Layout mirrors the reference's hot-path modules:
  zkhip.field            <- zkp/plonk/field.py        (FR, G1, G2, ec_mul/ec_add/ec_neg, roots of unity)
  zkhip.groth16.setup    <- zkp/groth16/setup.py      (sigma11..sigma22: fixed-base batches)
  zkhip.groth16.proving  <- zkp/groth16/proving.py    (proof_a/b/c: G1/G2 MSMs)
  zkhip.groth16.poly_utils <- zkp/groth16/poly_utils.py (hxr and the F_r helpers)
  zkhip.plonk.polynomial <- zkp/plonk/polynomial.py   (Polynomial, fft, ifft)
  zkhip.plonk.utils      <- zkp/plonk/utils.py        (coset_fft, coset_ifft)
  zkhip.plonk.kzg        <- zkp/plonk/kzg.py          (commit)
  zkhip.plonk.srs        <- zkp/plonk/srs.py          (SRS.generate)
All group and transform arithmetic runs in libzkhip.so on the GPU; there is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .field import FQ, FQ2, FR, G1, G2, Z1, CURVE_ORDER, ec_add, ec_mul, ec_neg  # noqa: F401

__all__ = ["FQ", "FQ2", "FR", "G1", "G2", "Z1", "CURVE_ORDER", "ec_add", "ec_mul", "ec_neg"]
