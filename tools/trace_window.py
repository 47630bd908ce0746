#!/usr/bin/env python3
"""Every kernel of the last `--window-ms` of a rocprofv3 --kernel-trace CSV in start order (start, end, queue, name, duration in us),
then per kernel name: launches, summed duration, and the time during which it was the ONLY kind of kernel on the chip -- what a
stage costs on the critical path of a prover call as opposed to what it overlaps with.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_x -- python3 tools/bench_groth16.py
    python3 tools/trace_window.py gpurun_out/prof_x --window-ms 10.5 [--anchor reduce_window] [--quiet]"""
import argparse, collections, csv, glob, os, re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--window-ms", type=float, default=11.0)
    ap.add_argument("--anchor", default="", help="end the window at the last kernel whose name contains this")
    ap.add_argument("--before-gap-ms", type=float, default=0.0, help="end the window where the last GPU pause longer than this begins")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    f = sorted(glob.glob(os.path.join(a.dir, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]

    def short(n):
        n = n.replace("(anonymous namespace)::", "").replace("void zk::", "").replace("zk::", "")
        g2 = "Fp2" in n
        n = re.sub(r"[<(].*", "", n).replace("_kernel", "").replace("msm_", "")
        return n + (":g2" if g2 else "")

    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in csv.DictReader(open(f)))
    anchored = [e for e in ev if a.anchor in e[2]] if a.anchor else ev
    tend = anchored[-1][1]
    if a.before_gap_ms:   # end at the last kernel before the LAST pause of the GPU longer than this (a tool's host-side checks after its timed proofs)
        last = ev[0][1]
        for s, e, n, q in ev:
            if s - last > a.before_gap_ms * 1e6:
                tend = last
            last = max(last, e)
    t0 = tend - int(a.window_ms * 1e6)
    win = [e for e in ev if t0 <= e[0] and e[1] <= tend + 1000]
    base = win[0][0]
    if not a.quiet:
        for s, e, n, q in win:
            print("%9.1f %9.1f  q%-3s %-28s %8.1f" % ((s - base) / 1e3, (e - base) / 1e3, q, n, (e - s) / 1e3))
    # sweep: time with exactly one distinct kernel name active
    pts = []
    for s, e, n, q in win:
        pts.append((s, 1, n))
        pts.append((e, -1, n))
    pts.sort()
    active = collections.Counter()
    alone = collections.Counter()
    busy = idle = 0
    last = pts[0][0]
    for t, d, n in pts:
        names = [k for k, v in active.items() if v > 0]
        if t > last:
            if len(names) == 1:
                alone[names[0]] += t - last
            if names:
                busy += t - last
            else:
                idle += t - last
        active[n] += d
        last = t
    tot = collections.Counter()
    cnt = collections.Counter()
    for s, e, n, q in win:
        tot[n] += e - s
        cnt[n] += 1
    print("%-30s %6s %10s %10s" % ("kernel", "n", "sum us", "alone us"))
    for n, v in tot.most_common():
        print("%-30s %6d %10.1f %10.1f" % (n, cnt[n], v / 1e3, alone[n] / 1e3))
    print("window %.2f ms: busy %.2f ms, idle %.2f ms, kernel-time sum %.2f ms, alone sum %.2f ms" %
          ((win[-1][1] - base) / 1e6, busy / 1e6, idle / 1e6, sum(tot.values()) / 1e6, sum(alone.values()) / 1e6))


if __name__ == "__main__":
    main()
