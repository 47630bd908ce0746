#!/usr/bin/env python3
"""Wall-clock of the reference-signature facade at the reference's own sizes (BASELINE.json configs[0]): the toy Groth16 proof
(hxr + proof_a / proof_b / proof_c of tests/groth16/conftest.py:39-56, W = 6, G = 4) and the toy PLONK proof (x^3 + x + 5 = 35, n = 4,
SRS of seed 42) through zkhip.groth16.* / zkhip.plonk.*, i.e. through the host-buffer C ABI (zk_msm_g1 / zk_msm_g2 / zk_ntt_fr /
zk_group_op), one call per primitive.  These proofs are launch- and copy-latency-bound; the numbers show what the drop-in path costs
per proof once the entry points keep their plans (zk_cache_stats), next to the same proof's first call.
    python tools/bench_facade.py --reps 5
Prints one JSON line; both proofs are checked (golden fixture / verifier)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))


def run(reps=5):
    from zkhip import _lib
    from zkhip.field import FQ, FQ2, FR
    from zkhip.groth16.poly_utils import getFRPoly1D, getFRPoly2D, getNumGates, getNumWires, ax_val, bx_val, cx_val, zx_val, hxr
    from zkhip.groth16.proving import proof_a, proof_b, proof_c
    from zkhip.groth16.setup import sigma11, sigma12, sigma14, sigma15, sigma21, sigma22
    from zkhip.plonk.circuit import Circuit
    from zkhip.plonk.preprocessor import preprocess
    from zkhip.plonk.prover import prove
    from zkhip.plonk.srs import SRS
    from zkhip.plonk.verifier import verify
    lib = _lib.load()
    _lib.check(lib.zk_cache_clear())
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "toy_groth16.json")))
    t = golden["inputs"]
    alpha, beta, gamma, delta, x_val = (FR(t[k]) for k in ("alpha", "beta", "gamma", "delta", "x_val"))
    Ax, Bx, Cx = getFRPoly2D(t["Ap"]), getFRPoly2D(t["Bp"]), getFRPoly2D(t["Cp"])
    Zx, Rx = getFRPoly1D(t["Z"]), getFRPoly1D(t["R"])
    numGates, numWires = getNumGates(Ax), getNumWires(Ax)
    Axv, Bxv, Cxv, Zxv = ax_val(Ax, x_val), bx_val(Bx, x_val), cx_val(Cx, x_val), zx_val(Zx, x_val)
    s11, s12 = sigma11(alpha, beta, delta), sigma12(numGates, x_val)
    s14 = sigma14(numWires, alpha, beta, delta, Axv, Bxv, Cxv, pub_r_indexs=t["pub"])
    s15 = sigma15(numGates, delta, x_val, Zxv)
    s21, s22 = sigma21(beta, delta, gamma), sigma22(numGates, x_val)
    r, s = FR(t["r"]), FR(t["s"])

    def groth16_prove():
        Hx, _ = hxr(Ax, Bx, Cx, Zx, t["R"])
        A = proof_a(s11, s12, Ax, Rx, r)
        B = proof_b(s21, s22, Bx, Rx, s)
        C = proof_c(s11, s12, s14, s15, Bx, Rx, Hx, s, r, A, pub_r_indexs=t["pub"])
        return A, B, C

    def timed(fn, n):
        out, ts = None, []
        for _ in range(n):
            t0 = time.perf_counter()
            out = fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return out, ts

    (A, B, C), first = timed(groth16_prove, 1)
    _, ts = timed(groth16_prove, reps)
    g1j = lambda v: (FQ(int(v[0])), FQ(int(v[1])))
    g2j = lambda v: (FQ2((int(v[0][0]), int(v[0][1]))), FQ2((int(v[1][0]), int(v[1][1]))))
    ok_g = A == g1j(golden["proof_A"]) and B == g2j(golden["proof_B"]) and C == g1j(golden["proof_C"])

    srs = SRS.generate(20, seed=42)
    circuit, a, b, c, pub = Circuit.x3_plus_x_plus_5_eq_35()
    pp = preprocess(circuit, srs)
    plonk_prove = lambda: prove(circuit, a, b, c, pub, pp, srs)
    proof, pfirst = timed(plonk_prove, 1)
    _, pts = timed(plonk_prove, reps)
    ok_p = bool(verify(proof, pub, pp, srs))
    return {"groth16_toy_prove_ms": round(min(ts), 3), "groth16_toy_prove_first_call_ms": round(first[0], 3), "groth16_equals_golden": bool(ok_g),
            "plonk_toy_prove_ms": round(min(pts), 3), "plonk_toy_prove_first_call_ms": round(pfirst[0], 3), "plonk_verifies": ok_p,
            "reps": reps, "plan_cache": _lib.cache_stats(),
            "note": "reference-signature facade over the host-buffer C ABI (configs[0] sizes): latency-bound, one library call per primitive"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    print(json.dumps(run(ap.parse_args().reps)))
