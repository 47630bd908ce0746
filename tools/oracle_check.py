#!/usr/bin/env python3
"""Expected points for the tools' closed-form checks, from the C oracle (oracle/bn254_oracle.c) -- the CHECKER only: nothing a
tool times or reports as throughput goes through here.  A tool that says "closed_form_ok" compares libzkhip's MSM result with
`(sum_i s_i * k_i mod r) * G` computed by these functions, so the expectation owes nothing to libzkhip's own group operations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import c_oracle  # noqa: E402
import py_ref  # noqa: E402


def g1_mul(k):
    """k * G1 as (x, y) integers | None."""
    return c_oracle.g1_mul(py_ref.G1, int(k) % py_ref.R)


def g2_mul(k):
    """k * G2 as ((x0, x1), (y0, y1)) integers | None."""
    return c_oracle.g2_mul(py_ref.G2, int(k) % py_ref.R)


def g1_ints(pt):
    """A facade G1 point (FQ, FQ) | None in the oracle's format."""
    return None if pt is None else (int(pt[0]), int(pt[1]))


def g2_ints(pt):
    """A facade G2 point (FQ2, FQ2) | None in the oracle's format."""
    return None if pt is None else tuple(tuple(int(c) for c in v.coeffs) for v in pt)


def msm_result_is(got, dot, g2=False):
    """True when the facade point `got` equals dot * G1 (or G2) by the oracle."""
    return (g2_ints(got) == g2_mul(dot)) if g2 else (g1_ints(got) == g1_mul(dot))
