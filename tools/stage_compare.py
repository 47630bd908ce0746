#!/usr/bin/env python3
"""Blocking (one MSM on the GPU at a time) stage times of the generic and the bound-bases MSM, side by side.

Under the pipelined bench the per-stage HIP-event times are inflated by the neighbouring lanes' kernels, so they
cannot say which stage of one mode is slower than the other's; this runs each MSM alone.

    python tools/stage_compare.py [--log-n 20] [--group g1|g2] [--reps 20]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--group", default="g1")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--n", type=int, default=0, help="points (default 2^log_n)")
    ap.add_argument("--bits", type=int, default=0, help="keep only the low BITS bits of every scalar (0 = uniform below r)")
    ap.add_argument("--witness-like", action="store_true", help="a quarter of the scalars 0, a quarter 1, the rest uniform (SURVEY.md section 8 row D2)")
    ap.add_argument("--lib", default="", help="another build of libzkhip.so (same-session A/B runs; default: the in-tree library)")
    args = ap.parse_args()
    import time

    import torch
    from zkhip.synthetic import random_scalars
    from zkhip import _lib
    if args.lib:
        _lib.LIB_PATH = args.lib
    from zkhip.device import MsmPlan

    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n = args.n or (1 << args.log_n)
    rng = np.random.default_rng(7)
    scalars = random_scalars(rng, n)
    ks = random_scalars(rng, n)
    if args.bits:
        for wd in range(4):
            keep = min(64, max(0, args.bits - 64 * wd))
            scalars[:, wd] &= np.uint64((1 << keep) - 1)
    if args.witness_like:
        pick = rng.random(n)
        scalars[pick < 0.25] = 0
        scalars[(pick >= 0.25) & (pick < 0.5)] = np.array([1, 0, 0, 0], dtype=np.uint64)
    if args.group == "g1":
        gen = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
        points = np.zeros((n, 8), dtype=np.uint64)
        _lib.check(lib.zk_fixed_base_g1(_lib.ptr(gen), _lib.ptr(ks), n, _lib.ptr(points)))
        group = _lib.GROUP_G1
    else:
        from zkhip.field import G2, g2_to_limbs
        gen = g2_to_limbs([G2])
        points = np.zeros((n, 16), dtype=np.uint64)
        _lib.check(lib.zk_fixed_base_g2(_lib.ptr(gen), _lib.ptr(ks), n, _lib.ptr(points)))
        group = _lib.GROUP_G2
    d_s = torch.from_numpy(scalars.view(np.int64)).to(dev)
    d_p = torch.from_numpy(points.view(np.int64)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    plan = MsmPlan(group, n)
    plan.set_profiling(True)

    def measure(pts):
        for _ in range(10):
            res = plan.run_limbs(d_s.data_ptr(), pts, n, st)
        acc = np.zeros(4)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            res = plan.run_limbs(d_s.data_ptr(), pts, n, st)
            acc += np.array(plan.stage_ms())
        wall = (time.perf_counter() - t0) / args.reps * 1e3
        return res, acc / args.reps, wall

    r0, s0, w0 = measure(d_p.data_ptr())
    plan.bind(d_p.data_ptr(), n, st)
    r1, s1, w1 = measure(None)
    names = ("prepare", "sort", "accumulate", "reduce")
    print("n = %d %s, blocking, ms per MSM" % (n, args.group))
    print("%-12s %9s %9s" % ("stage", "generic", "bound"))
    for k, nm in enumerate(names):
        print("%-12s %9.4f %9.4f" % (nm, s0[k], s1[k]))
    print("%-12s %9.4f %9.4f" % ("sum", s0.sum(), s1.sum()))
    print("%-12s %9.4f %9.4f" % ("wall", w0, w1))
    print("same result:", bool(np.array_equal(r0[0], r1[0]) and r0[1] == r1[1]))


if __name__ == "__main__":
    main()
