#!/usr/bin/env python3
"""Idle gaps of the GPU in a rocprofv3 kernel trace (CSV): merges all kernels of the last `--window-ms` before the last
kernel into busy intervals and lists the gaps -- shows host-side stalls (synchronous copies, stream synchronisations)
inside a prover call.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_x -- python3 tools/bench_groth16.py
    python3 tools/trace_gaps.py gpurun_out/prof_x --window-ms 11.5 [--anchor msm_reduce_block]"""
import argparse, csv, glob, os, re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--window-ms", type=float, default=12.0)
    ap.add_argument("--anchor", default="", help="end the window at the last kernel whose name contains this")
    ap.add_argument("--min-gap-us", type=float, default=40.0)
    a = ap.parse_args()
    f = sorted(glob.glob(os.path.join(a.dir, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"[<(].*", "", r["Kernel_Name"].replace("void zk::", "").replace("zk::", "")))
                for r in csv.DictReader(open(f)))
    anchored = [e for e in ev if a.anchor in e[2]] if a.anchor else ev
    tend = anchored[-1][1]
    t0 = tend - int(a.window_ms * 1e6)
    win = [e for e in ev if t0 <= e[0] <= tend]
    base, cur_e, idle = win[0][0], win[0][1], 0
    prev = win[0][2]
    for s, e, n in win[1:]:
        if s > cur_e:
            idle += s - cur_e
            if s - cur_e >= a.min_gap_us * 1e3:
                print("gap %9.1f -> %9.1f us (%6.0f us)   after %-28s before %s" % ((cur_e - base) / 1e3, (s - base) / 1e3, (s - cur_e) / 1e3, prev, n))
        if e > cur_e:
            cur_e, prev = e, n
    print("window %.2f ms, GPU idle %.2f ms" % ((cur_e - base) / 1e6, idle / 1e6))


if __name__ == "__main__":
    main()
