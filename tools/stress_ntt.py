#!/usr/bin/env python3
"""Randomised stress of the device transforms on one GPU: random sizes (one, two and three passes), directions, coset shifts, input
lengths (zero-padded), one to four jobs per call (zk_ntt_dev_multi), in place or out of place per job -- every output against the
oracle's transform of that job alone (oracle/: the checker, nothing else).
    python tools/stress_ntt.py --iters 200 [--seed 1] [--max-log 18]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-log", type=int, default=18)
    a = ap.parse_args()
    import torch
    import c_oracle as co
    import py_ref as o
    from zkhip.device import NttPlan
    from zkhip.synthetic import random_scalars
    rng = np.random.default_rng(a.seed)
    st = torch.cuda.current_stream().cuda_stream
    plans = {}
    bad = 0
    for it in range(a.iters):
        L = int(rng.integers(0, a.max_log + 1))
        n = 1 << L
        jobs = int(rng.integers(1, 5))
        inverse = bool(rng.integers(0, 2))
        k = [None, 5, int(rng.integers(2, 1 << 62))][int(rng.integers(0, 3))]
        in_len = n if rng.random() < 0.5 else int(rng.integers(1, n + 1))
        plan = plans.setdefault(L, NttPlan(L))
        w = o.get_root_of_unity(n)
        Xs = [random_scalars(rng, n) for _ in range(jobs)]
        want = []
        for X in Xs:
            Z = X.copy()
            Z[in_len:] = 0
            if k is not None and not inverse:
                Z = co.to_limbs([v * pow(k, i, o.R) % o.R for i, v in enumerate(co.from_limbs(Z))])
            Y = co.ntt_arr(Z, w, inverse)
            if k is not None and inverse:
                kinv = pow(k, -1, o.R)
                Y = co.to_limbs([v * pow(kinv, i, o.R) % o.R for i, v in enumerate(co.from_limbs(Y))])
            want.append(Y)
        d_in = [torch.from_numpy(X.view(np.int64).copy()).cuda() for X in Xs]
        d_out = [t if rng.random() < 0.5 else torch.full((n, 4), -1, dtype=torch.int64, device="cuda") for t in d_in]
        plan.run_multi([(x.data_ptr(), y.data_ptr()) for x, y in zip(d_in, d_out)], in_len, inverse, k, st)
        torch.cuda.synchronize()
        for b in range(jobs):
            if not np.array_equal(d_out[b].cpu().numpy().view(np.uint64), want[b]):
                bad += 1
                print("MISMATCH it %d L %d jobs %d job %d inverse %s k %s in_len %d" % (it, L, jobs, b, inverse, k, in_len), flush=True)
        if it % 25 == 24:
            print("iter %d ok so far" % (it + 1), flush=True)
    print("done: %d mismatches" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
