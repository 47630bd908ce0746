"""PLONK prover round 4 (mirrors zkp/plonk/prover/round4.py:40-79): challenge zeta; the six openings a(zeta), b(zeta), c(zeta),
S_sigma1(zeta), S_sigma2(zeta), z(zeta omega), appended to the transcript in that order."""


def execute(state):
    state.zeta = state.transcript.challenge_scalar(b"zeta")
    zeta, pp, pr = state.zeta, state.preprocessed, state.proof
    pr.a_eval = state.a_poly.evaluate(zeta)
    pr.b_eval = state.b_poly.evaluate(zeta)
    pr.c_eval = state.c_poly.evaluate(zeta)
    pr.s_sigma1_eval = pp.s_sigma1_poly.evaluate(zeta)
    pr.s_sigma2_eval = pp.s_sigma2_poly.evaluate(zeta)
    pr.z_omega_eval = state.z_poly.evaluate(zeta * state.omega)
    for name in ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval"):
        state.transcript.append_scalar(name.encode(), getattr(pr, name))
