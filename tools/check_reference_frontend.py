#!/usr/bin/env python3
"""Pins the toy Groth16 INPUTS to the reference itself -- the one part of the reference that runs in the build container.

Everything on the hot path imports py_ecc (absent: SURVEY.md section 8 C1), but the reference's front end does not:
`zkp.groth16.code_to_r1cs` (Python source -> flat code -> R1CS + witness, code_to_r1cs.py:156-242) and
`zkp.groth16.qap_creator_lcm` (R1CS -> integer QAP, qap_creator_lcm.py:98-135).  This script imports those two modules from
/root/reference, runs the toy program of the reference's own fixture (tests/groth16/conftest.py:39-56, 59-72: `y = x**3; return
y + x + 5` at x = 3), and

  * asserts that the witness, `Ap / Bp / Cp` and `Z` it produces equal what the oracle hard-codes (oracle/py_ref.TOY) and
    what tests/golden/toy_groth16.json carries as "inputs";
  * writes them, together with the R1CS matrices, to tests/golden/toy_frontend_reference.json as REFERENCE-DERIVED data
    (numbers only; no reference source travels).

Build container only: /root/reference does not exist on the GPU box.  tests/test_reference_frontend.py runs `derive()` when the
reference is present and compares it with the committed file; without it the test only checks the committed file against the
oracle and the golden fixture.

    python tools/check_reference_frontend.py            # check + (re)write the fixture
    python tools/check_reference_frontend.py --check    # check only
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
FIXTURE = os.path.join(ROOT, "tests", "golden", "toy_frontend_reference.json")

# the reference fixture's constants (tests/groth16/conftest.py:39-45); data, not code
TEST_CODE = "\ndef qeval(x):\n    y = x**3\n    return y + x + 5\n"
TEST_INPUT_VARS = [3]


def _as_int(v):
    """The front end works in floats (Lagrange interpolation times the Vandermonde determinant); the reference turns them into
    field elements with round() (zkp/groth16/poly_utils.py:75-79).  Refuse anything that is not an integer to 1e-6."""
    r = round(v)
    if abs(v - r) > 1e-6:
        raise ValueError("reference front end produced a non-integer coefficient: %r" % (v,))
    return int(r)


def derive():
    """Runs the reference's front end on the toy program -> dict of plain ints."""
    if not os.path.isdir(REFERENCE):
        raise FileNotFoundError(REFERENCE)
    sys.path.insert(0, REFERENCE)
    try:
        from zkp.groth16.code_to_r1cs import code_to_r1cs_with_inputs, initialize_symbol
        from zkp.groth16.qap_creator_lcm import r1cs_to_qap_times_lcm
    finally:
        sys.path.remove(REFERENCE)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):          # the front end prints its intermediate steps
        initialize_symbol()                                  # global symbol counter (code_to_r1cs.py:53-59)
        r, A, B, C = code_to_r1cs_with_inputs(TEST_CODE, TEST_INPUT_VARS)
        Ap, Bp, Cp, Z = r1cs_to_qap_times_lcm(A, B, C)
    ints = lambda m: [[_as_int(v) for v in row] for row in m]
    return {"R": [_as_int(v) for v in r], "r1cs_A": ints(A), "r1cs_B": ints(B), "r1cs_C": ints(C),
            "Ap": ints(Ap), "Bp": ints(Bp), "Cp": ints(Cp), "Z": [_as_int(v) for v in Z]}


def check_against_oracle(d):
    """The reference-derived values against oracle/py_ref.TOY and the committed golden inputs; raises AssertionError."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import py_ref
    for key in ("R", "Ap", "Bp", "Cp", "Z"):
        assert d[key] == py_ref.TOY[key], "oracle/py_ref.TOY[%r] differs from the reference front end" % key
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "toy_groth16.json")))["inputs"]
    for key in ("R", "Ap", "Bp", "Cp", "Z"):
        assert d[key] == golden[key], "tests/golden/toy_groth16.json inputs[%r] differs from the reference front end" % key
    # the R1CS the QAP came from is satisfied by the witness (tests/groth16/test_integration.py:11-21)
    dot = lambda row: sum(a * b for a, b in zip(row, d["R"]))
    for ra, rb, rc in zip(d["r1cs_A"], d["r1cs_B"], d["r1cs_C"]):
        assert dot(ra) * dot(rb) == dot(rc)
    # and the QAP vanishes on the domain {1..G} for that witness: A and B carry the Vandermonde determinant once, C twice
    # (qap_creator_lcm.py:110-127), so r.A(x) * r.B(x) - r.C(x) = 0 there as it stands (tests/groth16/test_integration.py:24-33)
    G = len(d["Ap"][0])
    ev = lambda poly, x: sum(c * x ** i for i, c in enumerate(poly))
    for x in range(1, G + 1):
        a = sum(w * ev(p, x) for w, p in zip(d["R"], d["Ap"]))
        b = sum(w * ev(p, x) for w, p in zip(d["R"], d["Bp"]))
        c = sum(w * ev(p, x) for w, p in zip(d["R"], d["Cp"]))
        assert a * b == c, x
        assert ev(d["Z"], x) == 0


def main():
    d = derive()
    check_against_oracle(d)
    out = {"source": "derived by importing /root/reference/zkp/groth16/{code_to_r1cs,qap_creator_lcm}.py (py_ecc-free) and running the toy "
                     "program of tests/groth16/conftest.py:39-45, 59-72; script: tools/check_reference_frontend.py",
           "program": TEST_CODE, "input_vars": TEST_INPUT_VARS, **d}
    if "--check" in sys.argv[1:]:
        committed = json.load(open(FIXTURE))
        assert {k: committed[k] for k in d} == d, "tests/golden/toy_frontend_reference.json is stale"
        print("reference front end == oracle/py_ref.TOY == committed fixtures")
        return
    with open(FIXTURE, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("reference front end == oracle/py_ref.TOY; wrote", os.path.relpath(FIXTURE, ROOT))


if __name__ == "__main__":
    main()
