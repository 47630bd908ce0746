#!/usr/bin/env python3
"""PLONK prove() wall-clock on a synthetic 2^log_n-gate circuit with everything resident in HBM
(zkhip.plonk.prover_device.DevicePlonk): alternating multiplication / addition gates, each output wired to the next
gate's left input.  The proof is checked by the verifier (two pairings on the host).
    python tools/bench_plonk.py --log-n 20 --reps 3"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))



def committed_timeline(name):
    """The summary line of the committed kernel timeline of one proof (profiles/*_<name>_timeline.txt, written from a
    rocprofv3 --kernel-trace run by tools/trace_window.py): GPU-busy time and the sum of the kernel durations inside the proof's
    window.  Quoted with its source, not measured by this run; None when no such file exists."""
    import glob, re
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_timeline.txt" % name)))
    if not found:
        return None
    m = None
    for ln in open(found[-1]):
        m = re.match(r"window ([0-9.]+) ms: busy ([0-9.]+) ms, idle ([0-9.]+) ms, kernel-time sum ([0-9.]+) ms", ln) or m
    if not m:
        return None
    return {"source": os.path.relpath(found[-1], ROOT), "window_ms": float(m.group(1)), "gpu_busy_ms": float(m.group(2)),
            "gpu_idle_ms": float(m.group(3)), "kernel_time_sum_ms": float(m.group(4)),
            "note": "from a separate rocprofv3 --kernel-trace run of this tool (kernels of concurrent MSM lanes overlap, so the sum may exceed the window)"}

def mulmod_limbs(vals):
    from zkhip import _lib
    return _lib.ints_to_limbs(vals)


def run(log_n, reps, profile=False, pause_after_proofs=0.0, warm=3):
    import torch
    from zkhip import _lib
    from zkhip.field import CURVE_ORDER as R, G1, G2, fixed_base_mul, get_root_of_unity
    from zkhip.plonk.prover_device import DevicePlonk
    from zkhip.plonk.srs import DeviceSRS
    from zkhip.plonk.verifier import verify
    n = 1 << log_n
    rng = np.random.default_rng(11)
    t0 = time.perf_counter()
    ys = [int(v) for v in rng.integers(1, 1 << 50, size=n)]
    a, c = [0] * n, [0] * n
    cur = 3
    for i in range(n):
        a[i] = cur
        cur = cur * ys[i] % R if i % 2 == 0 else (cur + ys[i]) % R
        c[i] = cur
    t_wit = time.perf_counter() - t0
    def preprocess():
        lib = _lib.load()
        from zkhip.device import FrVec
        st = torch.cuda.current_stream().cuda_stream
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
        two = _lib.ints_to_limbs([0, 1, R - 1])                                             # the values the selector columns take
        col = lambda even_val, odd_val: up(np.tile(np.stack([two[even_val], two[odd_val]]), (n // 2, 1)))
        sel = [col(0, 1), col(0, 1), col(2, 2), col(1, 0), col(0, 0)]                        # q_l, q_r, q_o, q_m, q_c
        # labels omega^i on the device (zk_fr_scale_powers_dev), the permutation columns as shifted multiples of them:
        # sigma: c_{i-1} <-> a_i swapped, everything else fixed
        fv = FrVec()
        ones = up(np.tile(two[1], (n + 8, 1)))
        dom = ones[:n].clone()
        fv.scale_powers(dom.data_ptr(), n, int(get_root_of_unity(n)), st)
        s1, s2, s3 = (torch.empty_like(dom) for _ in range(3))
        FrVec.lincomb(s1[1:].data_ptr(), [dom.data_ptr()], [3], n - 1, stream=st)           # a_i  -> position of c_{i-1}: 3 * omega^(i-1)
        s1[:1] = dom[:1]                                                                      # a_0  -> itself
        FrVec.lincomb(s2.data_ptr(), [dom.data_ptr()], [2], n, stream=st)                    # b_i  -> itself: 2 * omega^i
        s3[:n - 1] = dom[1:]                                                                  # c_i  -> position of a_{i+1}: omega^(i+1)
        FrVec.lincomb(s3[n - 1:].data_ptr(), [dom[n - 1:].data_ptr()], [3], 1, stream=st)    # c_{n-1} -> itself
        sig = [s1, s2, s3]
        # SRS: [tau^i]_1 built on the device (zkhip.plonk.srs.DeviceSRS: srs.py:77-85 with both loops on the GPU)
        tau = 0xC0FFEE1234567
        fv.close()
        P = DeviceSRS.generate(n + 7, tau=tau).d_g1
        dev = DevicePlonk(sel, sig, P)
        return dev, tau

    t0 = time.perf_counter()
    dev, tau = preprocess()
    torch.cuda.synchronize()
    t_pre_first = time.perf_counter() - t0              # includes what a process pays once (code objects, first launches)
    del dev
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    dev, tau = preprocess()
    torch.cuda.synchronize()
    t_pre = time.perf_counter() - t0
    cols = [torch.from_numpy(_lib.ints_to_limbs(v).view(np.int64)).cuda() for v in (a, ys, c)]
    # the witness above lives in Python lists of 2^20 ints: park those objects outside the collector, or a generation-2 sweep
    # over them lands in one of the timed calls (+20 ms in every run before this was done)
    import gc
    gc.collect()
    gc.freeze()
    times = []
    # (reps + 1 timed proofs after `warm` untimed ones: the set-up above leaves the chip idle for seconds, and the first proofs after it
    # run on clocks that are still coming back -- 17.6 17.0 16.9 16.75 ms for four proofs in a row; tools/bench_groth16.py does the same)
    for _ in range(warm):
        dev.prove(*cols)
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        proof = dev.prove(*cols)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)

    if pause_after_proofs:                          # a kernel trace can then be cut at this pause (tools/trace_window.py --before-gap-ms)
        time.sleep(pause_after_proofs)
    if profile:                                     # where the HOST time of one prove() goes (cumulative, top entries)
        import cProfile, pstats
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        pr.enable()
        proof = dev.prove(*cols)            # (the checks below read the LAST proof's polynomials off the device: keep them a pair)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)

    class Srs:
        g2_powers = [G2] + fixed_base_mul(G2, [tau])
    ok = verify(proof, [], dev.preprocessed(), Srs)
    # independent of the verifier and of the MSM: tau is known here, so each of the nine commitments of the proof (and the eight
    # of the preprocessing) must be p(tau) * G1 -- p(tau) by scale-and-sum on the device, the scalar multiplication by the C ORACLE
    # (checker only; nothing timed goes through it)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import py_ref
    mismatches = dev.closed_form_mismatches(proof, tau, mul=lambda k: c_oracle.g1_mul(py_ref.G1, int(k) % py_ref.R))
    return {"commitments_equal_p_of_tau_times_G1": not mismatches, "commitment_mismatches": mismatches, "expected_points_from": "oracle/bn254_oracle.c scalar multiplication",
            "log_n": log_n, "gates": n, "prove_ms": round(min(times[1:]) * 1e3, 3), "prove_ms_all": [round(t * 1e3, 3) for t in times[1:]],
            "first_call_ms": round(times[0] * 1e3, 3), "untimed_proofs_before": warm, "witness_gen_s_python": round(t_wit, 2), "preprocess_s": round(t_pre, 3), "preprocess_s_first_call_in_process": round(t_pre_first, 3), "verified": bool(ok), "kernel_trace_of_one_proof": committed_timeline("plonk")}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lib", default="", help="another build of libzkhip.so (same-session A/B runs)")
    ap.add_argument("--profile", action="store_true", help="cProfile one more prove() and print the host-side hot spots to stderr")
    ap.add_argument("--pause-after-proofs", type=float, default=0.0, help="seconds of GPU idleness between the timed proofs and the checks (marks the end of the last proof in a kernel trace)")
    args = ap.parse_args()
    if args.lib:
        from zkhip import _lib
        _lib.LIB_PATH = args.lib
    print(json.dumps(run(args.log_n, args.reps, args.profile, args.pause_after_proofs)), flush=True)
