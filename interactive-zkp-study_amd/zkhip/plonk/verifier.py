"""PLONK verifier (mirrors zkp/plonk/verifier.py:42-208): transcript replay, the linearised commitment
[D], the batched opening check  e(W_z + u W_zw, [tau]_2) == e(z W_z + u z w W_zw + [F] + u [z] - [E], [1]_2).
The ~20 scalar multiplications of the reference are folded into two G1 MSMs on the GPU backend; the
two pairings run on the host."""
from ..field import FR, G1, ec_neg, msm_g1, pairing_check
from .prover import linearisation_scalars
from .transcript import Transcript


def verify(proof, public_inputs, preprocessed, srs):
    pp, pr = preprocessed, proof
    n, omega = pp.n, pp.omega
    tr = Transcript()
    for name in ("a_comm", "b_comm", "c_comm"):
        tr.append_point(name.encode(), getattr(pr, name))
    beta = tr.challenge_scalar(b"beta")
    gamma = tr.challenge_scalar(b"gamma")
    tr.append_point(b"z_comm", pr.z_comm)
    alpha = tr.challenge_scalar(b"alpha")
    for name in ("t_lo_comm", "t_mid_comm", "t_hi_comm"):
        tr.append_point(name.encode(), getattr(pr, name))
    zeta = tr.challenge_scalar(b"zeta")
    for name in ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval"):
        tr.append_scalar(name.encode(), getattr(pr, name))
    v = tr.challenge_scalar(b"v")
    u = tr.challenge_scalar(b"u")

    a_e, b_e, c_e = FR(pr.a_eval), FR(pr.b_eval), FR(pr.c_eval)
    s1_e, s2_e, zw_e, r_e = FR(pr.s_sigma1_eval), FR(pr.s_sigma2_eval), FR(pr.z_omega_eval), FR(pr.r_eval)
    zh_zeta, l1_zeta, perm_z, perm_s3, r0 = linearisation_scalars(alpha, beta, gamma, zeta, n, omega, a_e, b_e, c_e, s1_e, s2_e, zw_e)
    zeta_n = zeta ** n
    v2, v3, v4, v5, v6 = v ** 2, v ** 3, v ** 4, v ** 5, v ** 6
    # e = t(zeta) + v r + v^2 a + v^3 b + v^4 c + v^5 s1 + v^6 s2 + u z_w   with t(zeta) = r / Z_H(zeta)
    e_scalar = r_e / zh_zeta + v * r_e + v2 * a_e + v3 * b_e + v4 * c_e + v5 * s1_e + v6 * s2_e + u * zw_e
    # B = zeta W_z + u zeta omega W_zw + [F] + u [z] - e G1, with
    # [F] = t_lo + zeta^n t_mid + zeta^2n t_hi + v [D] + v r0 G1 + v^2 a + v^3 b + v^4 c + v^5 s1 + v^6 s2
    # [D] = a b q_M + a q_L + b q_R + c q_O + q_C + (perm_z + alpha^2 L1) z - perm_s3 s_sigma3
    scalars = [zeta, u * zeta * omega, FR(1), zeta_n, zeta_n * zeta_n,
               v * a_e * b_e, v * a_e, v * b_e, v * c_e, v,
               v * (perm_z + alpha * alpha * l1_zeta) + u, FR(0) - v * perm_s3,
               v * r0 - e_scalar, v2, v3, v4, v5, v6]
    points = [pr.W_zeta_comm, pr.W_zeta_omega_comm, pr.t_lo_comm, pr.t_mid_comm, pr.t_hi_comm,
              pp.q_m_comm, pp.q_l_comm, pp.q_r_comm, pp.q_o_comm, pp.q_c_comm,
              pr.z_comm, pp.s_sigma3_comm,
              G1, pr.a_comm, pr.b_comm, pr.c_comm, pp.s_sigma1_comm, pp.s_sigma2_comm]
    B = msm_g1(scalars, points)
    A = msm_g1([FR(1), u], [pr.W_zeta_comm, pr.W_zeta_omega_comm])
    return pairing_check([(A, srs.g2_powers[1]), (ec_neg(B), srs.g2_powers[0])])
