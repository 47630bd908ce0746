#!/usr/bin/env python3
"""Randomised comparison of the device-resident PLONK prover with the list prover (same blinding): random gate mixes,
copy constraints and sizes; every proof field must match and verify.   python tools/stress_plonk.py --iters 12"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from zkhip import _lib
    from zkhip.field import FR, CURVE_ORDER as R, g1_to_limbs
    from zkhip.plonk.circuit import Circuit
    from zkhip.plonk.permutation import build_permutation_polynomials
    from zkhip.plonk.preprocessor import preprocess
    from zkhip.plonk.prover import Proof, prove
    from zkhip.plonk.prover_device import DevicePlonk
    from zkhip.plonk.srs import SRS
    from zkhip.plonk.verifier import verify
    rng = np.random.default_rng(a.seed)
    lim = lambda vals: _lib.ints_to_limbs([int(v) % R for v in vals])
    srs = SRS.generate(300, seed=int(rng.integers(1, 1 << 30)))
    bad = 0
    for it in range(a.iters):
        rows = int(rng.integers(3, 200))
        c = Circuit()
        av, bv, cv = [], [], []
        vals = []                                  # (gate, wire, value) of every output so far, for copy constraints
        for i in range(rows):
            kind = int(rng.integers(0, 3))
            if vals and rng.random() < 0.7:        # reuse an earlier output as the left input
                g0, w0, x = vals[int(rng.integers(0, len(vals)))]
            else:
                g0, x = None, int(rng.integers(0, 1 << 60))
            y = int(rng.integers(0, 1 << 60))
            if kind == 0:
                g = c.add_multiplication_gate(); out = x * y % R
            elif kind == 1:
                g = c.add_addition_gate(); out = (x + y) % R
            else:
                k = int(rng.integers(0, 1 << 30)); g = c.add_constant_gate(k); y = 0; out = (x + k) % R
            if g0 is not None:
                c.add_copy_constraint(g0, w0, g, 0)
            av.append(FR(x)); bv.append(FR(y)); cv.append(FR(out))
            vals.append((g, 2, out))
        pp = preprocess(c, srs)
        n = pp.n
        pad = lambda col: list(col) + [FR(0)] * (n - len(col))
        av, bv, cv = pad(av), pad(bv), pad(cv)
        blinding = [int(v) for v in rng.integers(1, 1 << 62, size=9)]
        want = prove(c, av, bv, cv, [], pp, srs, blinding=blinding)
        sel = [lim(col) for col in c.get_selector_polynomials()]
        sig = [lim(col) for col in build_permutation_polynomials(pp.sigma, n, pp.domain)]
        dev = DevicePlonk(sel, sig, g1_to_limbs(srs.g1_powers))
        got = dev.prove(lim(av), lim(bv), lim(cv), blinding=blinding)
        same = all(getattr(got, f) == getattr(want, f) for f in Proof.FIELDS)
        ok = same and verify(got, [], dev.preprocessed(), srs)
        if not ok:
            bad += 1
        print("iter", it, "rows", rows, "n", n, "identical" if same else "DIFFERENT", "verified" if ok else "FAILED", flush=True)
    print("done: %d failures" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
