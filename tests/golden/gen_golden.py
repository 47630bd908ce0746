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


if __name__ == "__main__":
    toy_groth16()
    kzg()
    ntt()
