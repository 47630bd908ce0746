"""PLONK prover round 1 (mirrors zkp/plonk/prover/round1.py:55-108): wire polynomials a, b, c by inverse NTT (GPU), degree-1
blinding (b1 + b2 x) Z_H, three commitments (G1 MSMs on the GPU), appended to the transcript.  PI(x) is hard-wired to zero, as
in the reference (round1.py:59-75)."""
from ..kzg import commit
from ..polynomial import Polynomial
from .common import pad_rows, times_vanishing


def execute(state):
    n, omega = state.n, state.omega
    state.pi_poly = Polynomial.zero()
    polys = []
    for vals in (state.a_vals, state.b_vals, state.c_vals):
        p = Polynomial.from_evaluations(pad_rows(vals, n), omega)
        polys.append(p + times_vanishing(state._blind(2), n))
    state.a_poly, state.b_poly, state.c_poly = polys
    for name, p in zip(("a_comm", "b_comm", "c_comm"), polys):
        comm = commit(p, state.srs)
        setattr(state.proof, name, comm)
        state.transcript.append_point(name.encode(), comm)
