#!/usr/bin/env python3
"""MSM (G1, optionally G2) and NTT throughput across sizes on one GPU, each result checked by a
size-independent property (MSM: closed form (sum s_i k_i)*G; NTT: inverse(forward(x)) == x).
    python tools/bench_sizes.py --msm 16,18,20,22,24 --ntt 16,18,20,22,24 [--g2 16,18,20]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
from zkhip.synthetic import random_scalars, limbs_dot_mod_r


from zkhip.synthetic import arithmetic_dot, arithmetic_points  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--msm", default="16,18,20,22")
    ap.add_argument("--g2", default="")
    ap.add_argument("--ntt", default="16,18,20,22,24")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--chunk-log", type=int, default=0, help="chunk size of MSMs beyond it (zk_msm_plan_create_ex; 0 = the library's 2^22)")
    args = ap.parse_args()
    import torch
    from zkhip import _lib
    from zkhip.device import MsmPlan, NttPlan
    from zkhip.field import G2, g2_to_limbs
    from oracle_check import msm_result_is
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    out = {"msm_g1": [], "msm_g2": [], "ntt": []}
    for group, sizes in (("g1", args.msm), ("g2", args.g2)):
        for L in [int(v) for v in sizes.split(",") if v]:
            n = 1 << L
            rng = np.random.default_rng(100 + L)
            big = group == "g1" and L >= 23     # closed form without 2^L Python integers
            S = random_scalars(rng, n)
            K = None if big else random_scalars(rng, n)
            if big:
                k0, dd = 0x1234567890ABCDEF >> 1, 0x9E3779B1
                P = arithmetic_points(lib, n, k0, dd)
                plan = MsmPlan(_lib.GROUP_G1, n, chunk_log=args.chunk_log)
            elif group == "g1":
                base = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
                P = np.zeros((n, 8), dtype=np.uint64)
                _lib.check(lib.zk_fixed_base_g1(_lib.ptr(base), _lib.ptr(K), n, _lib.ptr(P)))
                plan = MsmPlan(_lib.GROUP_G1, n, chunk_log=args.chunk_log)
            else:
                base = g2_to_limbs([G2])
                P = np.zeros((n, 16), dtype=np.uint64)
                _lib.check(lib.zk_fixed_base_g2(_lib.ptr(base), _lib.ptr(K), n, _lib.ptr(P)))
                plan = MsmPlan(_lib.GROUP_G2, n)
            dS, dP = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(P.view(np.int64)).cuda()
            plan.set_profiling(True)
            res = plan.run(dS.data_ptr(), dP.data_ptr(), n, st)
            ts = []
            for _ in range(args.reps):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                res = plan.run(dS.data_ptr(), dP.data_ptr(), n, st)
                ts.append(time.perf_counter() - t0)
            dot = arithmetic_dot(S, k0, dd) if big else limbs_dot_mod_r(S, K)
            ok = msm_result_is(res, dot, group == "g2")                 # expectation from the C oracle (tools/oracle_check.py)
            ms = min(ts) * 1e3
            rec = {"log_n": L, "ms": round(ms, 3), "points_per_s": n / (ms * 1e-3), "stage_ms": [round(v, 3) for v in plan.stage_ms()],
                   "window_bits": plan.window_bits(n), "closed_form_ok": bool(ok)}
            out["msm_" + group].append(rec)
            print(group, rec, flush=True)
            plan.close(); del dS, dP, plan
            torch.cuda.empty_cache()
    for L in [int(v) for v in args.ntt.split(",") if v]:
        n = 1 << L
        X = random_scalars(np.random.default_rng(200 + L), n)
        d = torch.from_numpy(X.view(np.int64)).cuda()
        ref = d.clone()
        plan = NttPlan(L)
        plan.run(d.data_ptr(), False, None, st); plan.run(d.data_ptr(), True, None, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            plan.run(d.data_ptr(), False, None, st); plan.run(d.data_ptr(), True, None, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (2 * args.reps)
        rec = {"log_n": L, "ms_per_transform": round(ms, 4), "elements_per_s": n / (ms * 1e-3), "algorithmic_GBps": 64.0 * n / (ms * 1e-3) / 1e9,
               "roundtrip_exact": bool(torch.equal(d, ref))}
        out["ntt"].append(rec)
        print("ntt", rec, flush=True)
        plan.close(); del d, ref
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
