"""PLONK prover round 2 (mirrors zkp/plonk/prover/round2.py:50-86): challenges beta, gamma; the permutation accumulator z
(zkp/plonk/permutation.py:89-137), interpolated by inverse NTT, degree-2 blinding, committed."""
from ..kzg import commit
from ..permutation import compute_accumulator
from ..polynomial import Polynomial
from .common import pad_rows, times_vanishing


def execute(state):
    state.beta = state.transcript.challenge_scalar(b"beta")
    state.gamma = state.transcript.challenge_scalar(b"gamma")
    n = state.n
    z_evals = compute_accumulator(pad_rows(state.a_vals, n), pad_rows(state.b_vals, n), pad_rows(state.c_vals, n),
                                  state.preprocessed.sigma, n, state.domain, state.beta, state.gamma)
    state.z_poly = Polynomial.from_evaluations(z_evals, state.omega) + times_vanishing(state._blind(3), n)
    state.proof.z_comm = commit(state.z_poly, state.srs)
    state.transcript.append_point(b"z_comm", state.proof.z_comm)
