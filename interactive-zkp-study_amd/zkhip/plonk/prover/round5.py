"""PLONK prover round 5 (mirrors zkp/plonk/prover/round5.py:46-175): challenge v; the linearisation polynomial r(x) and r(zeta),
the batched opening quotient W_zeta and the opening quotient W_zeta_omega of z, both committed."""
from ...field import FR
from ..kzg import commit
from ..polynomial import Polynomial, poly_div
from .common import linearisation_scalars


def execute(state):
    state.v = state.transcript.challenge_scalar(b"v")
    v, n, zeta, pp, pr = state.v, state.n, state.zeta, state.preprocessed, state.proof
    _, l1_zeta, perm_z, perm_s3, r0 = linearisation_scalars(state.alpha, state.beta, state.gamma, zeta, n, state.omega, pr.a_eval,
                                                           pr.b_eval, pr.c_eval, pr.s_sigma1_eval, pr.s_sigma2_eval, pr.z_omega_eval)
    pi_zeta = state.pi_poly.evaluate(zeta)
    r_poly = (pp.q_m_poly * (pr.a_eval * pr.b_eval) + pp.q_l_poly * pr.a_eval + pp.q_r_poly * pr.b_eval + pp.q_o_poly * pr.c_eval
              + pp.q_c_poly + state.z_poly * (perm_z + state.alpha * state.alpha * l1_zeta) - pp.s_sigma3_poly * perm_s3
              + Polynomial([pi_zeta + r0]))
    pr.r_eval = r_poly.evaluate(zeta)
    zeta_n = zeta ** n
    t_combined = state.t_lo_poly + state.t_mid_poly * zeta_n + state.t_hi_poly * (zeta_n * zeta_n)
    numer = t_combined - Polynomial([t_combined.evaluate(zeta)])
    v_pow = v
    for poly, value in ((r_poly, pr.r_eval), (state.a_poly, pr.a_eval), (state.b_poly, pr.b_eval), (state.c_poly, pr.c_eval),
                        (pp.s_sigma1_poly, pr.s_sigma1_eval), (pp.s_sigma2_poly, pr.s_sigma2_eval)):
        numer = numer + (poly - Polynomial([value])) * v_pow
        v_pow = v_pow * v
    w_zeta, _ = poly_div(numer, Polynomial([FR(0) - zeta, FR(1)]))
    w_zeta_omega, _ = poly_div(state.z_poly - Polynomial([pr.z_omega_eval]), Polynomial([FR(0) - zeta * state.omega, FR(1)]))
    pr.W_zeta_comm = commit(w_zeta, state.srs)
    pr.W_zeta_omega_comm = commit(w_zeta_omega, state.srs)
