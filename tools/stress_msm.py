#!/usr/bin/env python3
"""Randomised stress of the MSM pipeline on one GPU: random sizes (including chunked ones through the test knob),
scalar patterns, in-flight depths and the bound-bases mode, every result checked against the closed form (sum s_i k_i) * G.
    python tools/stress_msm.py --iters 60 [--seed 1]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-log", type=int, default=19)
    a = ap.parse_args()
    import torch
    from zkhip.synthetic import random_scalars, limbs_dot_mod_r
    from zkhip import _lib
    from zkhip.device import MsmPlan
    from zkhip.field import G2, g2_to_limbs, limbs_to_g1, limbs_to_g2
    from oracle_check import msm_result_is
    lib = _lib.load()
    rng = np.random.default_rng(a.seed)
    nmax = 1 << a.max_log
    K = random_scalars(rng, nmax)
    g1 = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
    P1 = np.zeros((nmax, 8), dtype=np.uint64)
    _lib.check(lib.zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(K), nmax, _lib.ptr(P1)))
    n2max = 1 << min(a.max_log, 15)
    P2 = np.zeros((n2max, 16), dtype=np.uint64)
    _lib.check(lib.zk_fixed_base_g2(_lib.ptr(g2_to_limbs([G2])), _lib.ptr(K[:n2max]), n2max, _lib.ptr(P2)))
    dP1, dP2 = torch.from_numpy(P1.view(np.int64)).cuda(), torch.from_numpy(P2.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    for it in range(a.iters):
        g2 = rng.random() < 0.25
        chunk = int(rng.choice([0, 12, 14, 18]))
        limit = 1 << (chunk or 22)
        cap = n2max if g2 else nmax
        n = int(min(cap, max(1, int(2 ** rng.uniform(0, np.log2(cap))) + int(rng.integers(0, 3)))))
        if not g2 and rng.random() < 0.3:
            n = int(rng.integers((1 << 17) + 1, cap + 1))          # the 16-bit-window / bound-bases range more often
        plan = MsmPlan(_lib.GROUP_G2 if g2 else _lib.GROUP_G1, n, chunk_log=chunk)
        bound = n > (1 << 17) and chunk in (0, 18) and rng.random() < 0.6
        if bound:
            plan.bind((dP2 if g2 else dP1).data_ptr(), n, st)
        depth = int(rng.integers(1, plan.max_in_flight() + 1))
        jobs = []
        for j in range(int(rng.integers(1, 5))):
            m = n if rng.random() < 0.5 else int(rng.integers(0, n + 1))
            S = random_scalars(rng, max(m, 1))[:m]
            pat = rng.integers(0, 5)
            if m and pat == 1:
                S[rng.random(m) < 0.5] = np.array([1, 0, 0, 0], dtype=np.uint64)
            elif m and pat == 2:
                S[:] = S[0]
            elif m and pat == 3:
                S[rng.random(m) < 0.3] = 0
            elif m and pat == 4:               # a handful of distinct scalars: several heavy buckets in every window at once
                S = S[rng.integers(0, min(m, int(rng.integers(2, 65))), size=m)]
            jobs.append((m, S, torch.from_numpy(np.ascontiguousarray(S).view(np.int64)).cuda()))
        pend, res = [], []
        for m, S, dS in jobs:
            if m > limit:                       # a chunked MSM takes all lanes: drain first, run it alone
                res += [plan.collect_limbs(t) for t in pend]
                pend = []
                res.append(plan.collect_limbs(plan.submit(dS.data_ptr(), None if bound else (dP2 if g2 else dP1).data_ptr(), m, st)))
                continue
            if len(pend) == depth:
                res.append(plan.collect_limbs(pend.pop(0)))
            pend.append(plan.submit(dS.data_ptr(), None if (bound and rng.random() < 0.8) else (dP2 if g2 else dP1).data_ptr(), m, st))
        res += [plan.collect_limbs(t) for t in pend]
        for (m, S, _), (limbs, inf) in zip(jobs, res):
            dot = limbs_dot_mod_r(S, K[:m]) if m else 0
            got = None if inf else (limbs_to_g2(limbs) if g2 else limbs_to_g1(limbs))[0]
            if not msm_result_is(got, dot, g2):                        # expectation from the C oracle (tools/oracle_check.py)
                bad += 1
                print("MISMATCH", it, "g2" if g2 else "g1", "n", n, "m", m, "chunk", chunk, "depth", depth, "bound", bound, flush=True)
        plan.close()
        if it % 10 == 9:
            print("iter", it + 1, "ok so far" if not bad else "FAILURES %d" % bad, flush=True)
    print("done: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
