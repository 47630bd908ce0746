"""CPU test: the toy Groth16 inputs are pinned to the reference's own front end.

tests/golden/toy_frontend_reference.json was produced by tools/check_reference_frontend.py, which imports the py_ecc-free part of
the reference (zkp/groth16/code_to_r1cs.py, qap_creator_lcm.py) in the build container and runs the toy program of
tests/groth16/conftest.py:39-72.  Here: the committed file against the oracle's hard-coded TOY and the golden fixture (always), and
-- where /root/reference is present (the build container; never the GPU box) -- against a fresh run of the reference itself."""
import importlib.util
import json
import os

import pytest

import py_ref as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_reference_frontend", os.path.join(ROOT, "tools", "check_reference_frontend.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_reference_inputs_equal_oracle_and_golden(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "toy_frontend_reference.json")))
    _tool().check_against_oracle(d)          # TOY, the golden "inputs", R1CS satisfied, QAP vanishes on {1..G}
    assert d["R"] == [1, 3, 35, 9, 27, 30]   # EXPECTED_R of tests/groth16/conftest.py:45
    # the oracle's whole toy pipeline starts from exactly these arrays
    t = o.toy_groth16()
    assert t["Ax"] == [[v % o.R for v in row] for row in d["Ap"]] and t["Zx"] == [v % o.R for v in d["Z"]]
    assert t["Rx"] == d["R"]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_reference_front_end_reproduces_the_committed_inputs(golden_dir):
    tool = _tool()
    d = tool.derive()
    committed = json.load(open(os.path.join(golden_dir, "toy_frontend_reference.json")))
    assert {k: committed[k] for k in d} == d
    tool.check_against_oracle(d)
