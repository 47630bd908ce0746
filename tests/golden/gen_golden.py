#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from oracle/py_ref.py.

Provenance: the reference's hot-path modules cannot be imported here (py_ecc is not installed;
ordinary ModuleNotFoundError), so these vectors come from this repo's own
pure-Python restatement of the reference's algorithm, cross-checked inside this script by a
second independent method (closed-form scalars, Horner evaluation, and the reference's own
F_r known-answers zkp/groth16/backend.py:355,363).  Inputs are the reference's own fixtures:
tests/groth16/conftest.py:39-56 (toy circuit + toxic waste), tests/plonk/test_crypto.py:28
(SRS seed 42), tests/plonk/test_foundation.py:491-508 (fft inputs).  Values are decimal strings in the
reference's wire style (plonk_serializers.py:23-68).

Run:  python tests/golden/gen_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import py_ref as o  # noqa: E402
import plonk_ref as pl  # noqa: E402


def s(v):
    return str(int(v))


def g1(pt):
    return None if pt is None else [s(pt[0]), s(pt[1])]


def g2(pt):
    return None if pt is None else [[s(pt[0][0]), s(pt[0][1])], [s(pt[1][0]), s(pt[1][1])]]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")
    print("wrote", name)


def toy_groth16():
    d = o.toy_groth16()
    # second method: closed-form scalars (zkp/groth16/test.py:303-325)
    assert d["proof_A"] == o.g1_multiply(o.G1, d["A"])
    assert d["proof_B"] == o.g2_multiply(o.G2, d["B"])
    assert d["proof_C"] == o.g1_multiply(o.G1, d["C"])
    # reference comment KATs (zkp/groth16/backend.py:355,363)
    assert d["A"] * d["B"] % o.R == 21888242871839275222246405745257275088548364400416033032405666501928354297837
    assert d["VAL"][0] == 17858330771234736835653075572704017103548042849750409710240473560856989375368
    t = o.TOY
    dump("toy_groth16.json", {
        "source": "tests/groth16/conftest.py:39-56; QAP = zkp/groth16/backend.py:85-110 with the live unscaled Z",
        "inputs": {k: t[k] for k in ("R", "alpha", "beta", "gamma", "delta", "x_val", "r", "s", "pub", "Ap", "Bp", "Cp", "Z")},
        "Hx": [s(v) for v in d["Hx"]], "remainder": [s(v) for v in d["rem"]],
        "VAL": [s(v) for v in d["VAL"]],
        "Ax_val": [s(v) for v in d["Ax_val"]], "Bx_val": [s(v) for v in d["Bx_val"]],
        "Cx_val": [s(v) for v in d["Cx_val"]], "Zx_val": s(d["Zx_val"]),
        "A": s(d["A"]), "B": s(d["B"]), "C": s(d["C"]),
        "sigma1_1": [g1(p) for p in d["s11"]], "sigma1_2": [g1(p) for p in d["s12"]],
        "sigma1_3": [g1(p) for p in d["s13"]], "sigma1_4": [g1(p) for p in d["s14"]],
        "sigma1_5": [g1(p) for p in d["s15"]],
        "sigma2_1": [g2(p) for p in d["s21"]], "sigma2_2": [g2(p) for p in d["s22"]],
        "proof_A": g1(d["proof_A"]), "proof_B": g2(d["proof_B"]), "proof_C": g1(d["proof_C"]),
    })


def kzg():
    tau = o.srs_tau(42)
    g1p, g2p = o.srs_generate(8, 42)
    polys = {"const7": [7], "lin_3_5": [3, 5], "one_plus_2x": [1, 2], "quad_1_1_1": [1, 1, 1],
             "deg8": [0] * 8 + [1], "zero": [0], "with_zero_coeffs": [5, 0, 0, 9, 0, 2]}
    commits = {}
    for name, c in polys.items():
        cm = o.kzg_commit(c, g1p)
        # second method: p(tau) * G1
        assert cm == o.g1_multiply(o.G1, o.horner(c, tau))
        commits[name] = {"coeffs": [s(v) for v in c], "commitment": g1(cm)}
    dump("kzg_seed42.json", {
        "source": "SRS.generate(8, seed=42): tests/plonk/test_crypto.py:28; commit cases: tests/plonk/test_crypto.py:113-159",
        "tau": s(tau), "g1_powers": [g1(p) for p in g1p], "g2_powers": [g2(p) for p in g2p], "commits": commits,
        "two_G1": g1(o.g1_double(o.G1)),
    })


def ntt():
    cases = {}
    for name, coeffs in {"n1": [7], "n4_1234": [1, 2, 3, 4], "n8_iota": list(range(8)),
                         "n8_3i_plus_1": [3 * i + 1 for i in range(8)],
                         "n16_mixed": [(i * i * 7919 + 13) % o.R for i in range(16)],
                         "n64_big": [pow(3, i * 97 + 5, o.R) for i in range(64)]}.items():
        n = len(coeffs)
        w = o.get_root_of_unity(n)
        ev = o.fft(coeffs, w)
        # second method: Horner evaluation at omega^i (tests/plonk/test_foundation.py:501-508)
        assert all(ev[i] == o.horner(coeffs, pow(w, i, o.R)) for i in range(n))
        assert o.ifft(ev, w) == [c % o.R for c in coeffs]
        cev = o.coset_fft(coeffs, w)
        assert all(cev[i] == o.horner(coeffs, 5 * pow(w, i, o.R) % o.R) for i in range(n))
        assert o.coset_ifft(cev, w) == [c % o.R for c in coeffs]
        cases[name] = {"coeffs": [s(c) for c in coeffs], "omega": s(w), "fft": [s(v) for v in ev],
                       "ifft_of_coeffs": [s(v) for v in o.ifft(coeffs, w)], "coset_fft_k5": [s(v) for v in cev]}
    dump("ntt_small.json", {
        "source": "fft inputs: tests/plonk/test_foundation.py:491-531; omega = get_root_of_unity(n), zkp/plonk/field.py:145-182",
        "omega_4": s(o.get_root_of_unity(4)), "cases": cases,
    })


PLONK_CASES = {
    # the reference's fixture: Circuit.x3_plus_x_plus_5_eq_35(), SRS.generate(20, seed=42) (tests/plonk/test_prover.py:43-60,
    # tests/plonk/test_e2e.py:52-66); blinding scalars are what secrets.randbelow would have drawn, fixed here
    "toy_x3": {"make": pl.circuit_x3_plus_x_plus_5_eq_35, "srs_degree": 20, "seed": 42,
               "blinding": [pow(7, 50 + k, o.R) for k in range(9)]},
    # a second circuit: six gates padded to n = 8, blinding with a zero top coefficient (Polynomial._trim shortens the blinder)
    "six_gates_n8": {"make": pl.circuit_six_gates, "srs_degree": 40, "seed": 42,
                     "blinding": [pow(11, 90 + k, o.R) for k in range(8)] + [0]},
}


def plonk():
    out = {"source": "zkp/plonk/prover/round1..5.py, verifier.py, preprocessor.py restated in oracle/plonk_ref.py (O(n^2) products, poly_div, "
                     "Horner); circuit / SRS fixtures: tests/plonk/test_prover.py:43-60; the nine blinding scalars are injected in draw order "
                     "(a a b b c c z z z)", "cases": {}}
    for name, spec in PLONK_CASES.items():
        srs = o.srs_generate(spec["srs_degree"], spec["seed"])
        circuit, a, b, c, pub = spec["make"]()
        gates_in = [list(g) for g in circuit.gates]
        pp = pl.preprocess(circuit, srs)
        assert pl.gates_satisfied(circuit, a, b, c)
        proof, st = pl.prove(circuit, a, b, c, pub, pp, srs, spec["blinding"], return_state=True)
        assert st.rem_zeta == [0] and st.rem_zeta_omega == [0]
        assert pl.verify(proof, pub, pp, srs)
        # second method for every field: commitments are p(tau) * G1, evaluations are Horner values of the stored polynomials,
        # the quotient satisfies C = t * Z_H at an independent point
        tau = o.srs_tau(spec["seed"])
        for fld, poly in (("a_comm", st.a_poly), ("b_comm", st.b_poly), ("c_comm", st.c_poly), ("z_comm", st.z_poly),
                          ("t_lo_comm", st.t_lo_poly), ("t_mid_comm", st.t_mid_poly), ("t_hi_comm", st.t_hi_poly)):
            assert getattr(proof, fld) == o.g1_multiply(o.G1, o.horner(poly, tau)), fld
        n, w = pp.n, pp.omega
        x = 0x1234567
        ev = lambda poly, at=x: o.horner(poly, at)
        zh = (pow(x, n, o.R) - 1) % o.R
        t_x = (ev(st.t_lo_poly) + pow(x, n, o.R) * ev(st.t_mid_poly) + pow(x, 2 * n, o.R) * ev(st.t_hi_poly)) % o.R
        aa, bb, cc, zz, zw = ev(st.a_poly), ev(st.b_poly), ev(st.c_poly), ev(st.z_poly), ev(st.z_poly, x * w % o.R)
        gate = (ev(pp.q_l_poly) * aa + ev(pp.q_r_poly) * bb + ev(pp.q_o_poly) * cc + ev(pp.q_m_poly) * aa * bb + ev(pp.q_c_poly)) % o.R
        num = (aa + st.beta * x + st.gamma) * (bb + st.beta * 2 * x + st.gamma) * (cc + st.beta * 3 * x + st.gamma) * zz % o.R
        den = ((aa + st.beta * ev(pp.s_sigma1_poly) + st.gamma) * (bb + st.beta * ev(pp.s_sigma2_poly) + st.gamma)
               * (cc + st.beta * ev(pp.s_sigma3_poly) + st.gamma) * zw) % o.R
        l1 = pl.lagrange_basis_eval(0, n, w, x)
        assert (gate + st.alpha * (num - den) + st.alpha * st.alpha * (zz - 1) * l1 - t_x * zh) % o.R == 0
        assert proof.r_eval == o.horner(st.r_poly, st.zeta)
        assert st.t_eval * pl.vanishing_poly_eval(n, st.zeta) % o.R == proof.r_eval     # the verifier's t(zeta) = r(zeta) / Z_H(zeta)
        out["cases"][name] = {
            "srs": {"max_degree": spec["srs_degree"], "seed": spec["seed"]},
            "gates": [[s(v) for v in g] for g in gates_in], "copy_constraints": [list(cc_) for cc_ in circuit.copy_constraints],
            "n": pp.n, "a_vals": [s(v) for v in a], "b_vals": [s(v) for v in b], "c_vals": [s(v) for v in c],
            "public_inputs": [s(v) for v in pub], "blinding": [s(v) for v in spec["blinding"]],
            "sigma": pp.sigma,
            "preprocessed": {k: g1(getattr(pp, k)) for k in ("q_l_comm", "q_r_comm", "q_o_comm", "q_m_comm", "q_c_comm",
                                                             "s_sigma1_comm", "s_sigma2_comm", "s_sigma3_comm")},
            "challenges": {k: s(getattr(st, k)) for k in ("beta", "gamma", "alpha", "zeta", "v")},
            "polys": {k: [s(v) for v in getattr(st, k)] for k in ("a_poly", "b_poly", "c_poly", "z_poly", "t_lo_poly", "t_mid_poly", "t_hi_poly",
                                                                   "r_poly")},
            "proof": {f: (g1(getattr(proof, f)) if f in pl.PROOF_POINTS else s(getattr(proof, f))) for f in pl.PROOF_FIELDS},
        }
    dump("plonk_proofs.json", out)


if __name__ == "__main__":
    toy_groth16()
    kzg()
    ntt()
    plonk()
