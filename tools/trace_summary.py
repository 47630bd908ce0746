#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace CSV, optionally only the launches after the first `--skip-frac` of the
time range (to drop warm-up):  python tools/trace_summary.py <kernel_trace.csv> [--skip-frac 0.5]"""
import argparse, csv, collections

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--skip-frac", type=float, default=0.0)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t0 + a.skip_frac * (t1 - t0)
tot, cnt = collections.Counter(), collections.Counter()
for r in rows:
    if int(r["Start_Timestamp"]) < cut:
        continue
    name = r["Kernel_Name"].split("(")[0][:70]
    tot[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[name] += 1
total = sum(tot.values())
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'%':>6s}")
for k, v in tot.most_common(40):
    print(f"{k:70s} {cnt[k]:7d} {v / 1e6:10.3f} {v / cnt[k] / 1e3:10.1f} {100 * v / total:6.1f}")
