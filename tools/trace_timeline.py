#!/usr/bin/env python3
"""Kernel timeline of the last pipelined MSM steps in a rocprofv3 --kernel-trace CSV: start, end, queue, kernel, duration (us), plus the
gaps between consecutive accumulate kernels.   python tools/trace_timeline.py <dir with *_kernel_trace.csv> [--steps 8]"""
import argparse, csv, glob, os, re

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--quiet", action="store_true")
ap.add_argument("--skip-last", type=int, default=0, help="ignore that many accumulate kernels at the end of the trace (blocking calls after the pipelined loop)")
a = ap.parse_args()
f = sorted(glob.glob(os.path.join(a.dir, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"[<(].*", "", r["Kernel_Name"].replace("void zk::", "").replace("zk::", "")).replace("msm_", ""),
             r["Queue_Id"]) for r in csv.DictReader(open(f)))
acc = [e for e in ev if e[2] == "accumulate_kernel" and e[1] - e[0] > 500000]
acc = acc[:len(acc) - a.skip_last] if a.skip_last else acc
acc = acc[-a.steps:]
t0, t1 = acc[0][0] - 200000, acc[-1][1] + 400000
if not a.quiet:
    for s, e, n, q in ev:
        if t0 <= s <= t1:
            print("%8.1f %8.1f  q%s  %-22s %7.1f" % ((s - t0) / 1e3, (e - t0) / 1e3, q, n, (e - s) / 1e3))
gaps = [(acc[i + 1][0] - acc[i][1]) / 1e3 for i in range(len(acc) - 1)]
print("accumulate durations us:", [round((e - s) / 1e3) for s, e, _, _ in acc])
print("gaps between accumulate kernels us:", [round(g) for g in gaps], "mean %.0f" % (sum(gaps) / len(gaps)))
print("period us: %.0f" % ((acc[-1][0] - acc[0][0]) / 1e3 / (len(acc) - 1)))
