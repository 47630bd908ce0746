#!/usr/bin/env python3
"""Collect the measurement artifacts kept under profiles/ (run from the repo root on the GPU box).

    python3 tools/collect_profiles.py run  [--round r02]   # on the GPU box: bench line + rocprofv3 passes -> gpurun_out/prof_<round>/
    python3 tools/collect_profiles.py fold [--round r02]   # anywhere: gpurun_out/prof_<round>/ -> profiles/<round>_*

`run` starts each program directly after `rocprofv3 ... --` (no shell/env hop) and keeps the
counter passes (`--pmc`) separate from the kernel trace, one counter per pass."""
import argparse, csv, glob, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SQ_COUNTERS = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"]
BENCH_QUICK = ["python3", "bench.py", "--cpu-sample", "0", "--groth16-log-m", "0", "--plonk-log-n", "0", "--no-witness-like", "--no-bound", "--no-g2", "--no-facade", "--sizes", "", "--sizes-ntt", ""]


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_hash", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    return bench


def sh(cmd, log):
    with open(log, "w") as f:
        rc = subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, cwd=ROOT)
    print("rc=%d  %s" % (rc, " ".join(cmd)), flush=True)
    return rc


def run(rnd):
    out = os.path.join(ROOT, "gpurun_out", "prof_" + rnd)
    os.makedirs(out, exist_ok=True)
    env_tmp = os.environ.setdefault("TMPDIR", "/tmp")
    # the arithmetic sources the counters of this run belong to: stamped NOW, on the tree that is being measured (fold only copies it)
    with open(os.path.join(out, "arithmetic_source_sha256.json"), "w") as f:
        json.dump({"arithmetic_source_sha256": _bench_module().arithmetic_source_hash()}, f)
    rc = sh(["python3", "bench.py"], os.path.join(out, "bench_line.log"))
    if rc:
        return rc
    rc = sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(out, "trace"), "--"] + BENCH_QUICK + ["--steps", "20", "--warmup", "4"],
            os.path.join(out, "trace.log"))
    if rc:
        return rc
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        rc = sh(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(out, "pmc_" + ctr), "--"] + BENCH_QUICK + ["--steps", "3", "--warmup", "1"],
                os.path.join(out, "pmc_%s.log" % ctr))
        if rc:
            return rc
    # issue-slot accounting of the shader engines (one pass: 7 SQ counters + GRBM_GUI_ACTIVE)
    return sh(["rocprofv3", "--pmc"] + SQ_COUNTERS + ["--output-format", "csv", "-d", os.path.join(out, "pmc_SQ"), "--"] + BENCH_QUICK + ["--steps", "3", "--warmup", "1"],
              os.path.join(out, "pmc_SQ.log"))


def fold_sq(src, dst, rnd):
    """profiles/<round>_pmc_sq_summary.csv: per kernel, the SQ counters averaged over its dispatches and what they say about
    the vector ALU.  SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles summed over all wavefronts (MI355X_MICROARCH.md),
    SQ_INSTS_VALU counts wavefront-instructions, GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
    found = sorted(glob.glob(os.path.join(src, "pmc_SQ", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not found:
        return
    acc, grid = {}, {}
    for r in csv.DictReader(open(found[-1])):
        k = short_name(r["Kernel_Name"])
        d = acc.setdefault(k, {}).setdefault(r["Dispatch_Id"], {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        grid.setdefault(k, {})[r["Dispatch_Id"]] = int(r.get("Grid_Size", 0) or 0)
    for k in acc:                                   # the headline workload's launches only: those with the kernel's largest grid
        top = max(grid[k].values())
        acc[k] = {i: v for i, v in acc[k].items() if grid[k][i] == top}
    # the sources the counters belong to (stamped by `run` on the measured tree): bench.py quotes the issue rate only while they are
    # unchanged; a collection without the stamp stays untagged, i.e. unquoted
    try:
        tag = json.load(open(os.path.join(src, "arithmetic_source_sha256.json"))).get("arithmetic_source_sha256")
    except (OSError, ValueError):
        tag = None
    with open(os.path.join(dst, rnd + "_pmc_sq_summary.meta.json"), "w") as f:
        json.dump({"arithmetic_source_sha256": tag}, f, indent=1)
    with open(os.path.join(dst, rnd + "_pmc_sq_summary.csv"), "w") as f:
        f.write("kernel,dispatches," + ",".join(SQ_COUNTERS) + ",valu_insts_per_simd_cycle,active_valu_frac_of_wave_cycles,wait_inst_frac_of_wave_cycles\n")
        for k, disp in sorted(acc.items(), key=lambda kv: -sum(d.get("SQ_BUSY_CYCLES", 0) for d in kv[1].values())):
            n = len(disp)
            avg = {c: sum(d.get(c, 0.0) for d in disp.values()) / n for c in SQ_COUNTERS}
            cyc = avg["GRBM_GUI_ACTIVE"] / 8.0                       # shader-clock cycles the dispatch was resident
            simd_cyc = cyc * 256 * 4
            per = avg["SQ_INSTS_VALU"] / simd_cyc if simd_cyc else 0.0
            wc = avg["SQ_WAVE_CYCLES"] or 1.0
            f.write('"%s",%d,%s,%.4f,%.4f,%.4f\n' % (k, n, ",".join("%.0f" % avg[c] for c in SQ_COUNTERS), per, avg["SQ_ACTIVE_INST_VALU"] / wc, avg["SQ_WAIT_INST_ANY"] / wc))


def short_name(full):
    m = re.match(r"(?:void )?(?:zk::)?([A-Za-z0-9_]+(?:<.*?>)?)\(", full)
    return m.group(1) if m else full


def fold(rnd):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + rnd)
    dst = os.path.join(ROOT, "profiles")
    line = [l for l in open(os.path.join(src, "bench_line.log")) if l.startswith("{")][-1]
    with open(os.path.join(dst, rnd + "_bench_line.json"), "w") as f:
        f.write(line)
    stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if stats:
        with open(stats[-1]) as f, open(os.path.join(dst, rnd + "_bench_kernel_stats.csv"), "w") as g:
            g.write(f.read())
    # The summary's average mixes in the starts of the priming / warm-up / timed sequences, where three lanes enter their
    # accumulate kernels together and each launch lasts twice as long; the launches of the TIMED steps are the last ones.
    traces = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if traces:
        rows_t = [r for r in csv.DictReader(open(traces[-1])) if "msm_accumulate_kernel" in r["Kernel_Name"] and "Fp2" not in r["Kernel_Name"]]
        gsz = lambda r: int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        top = max(gsz(r) for r in rows_t)
        d = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows_t if gsz(r) == top)
        timed = [x[1] for x in d[-20:]]
        bl = json.loads(line)
        with open(os.path.join(dst, rnd + "_accumulate_launches.json"), "w") as f:
            json.dump({"kernel": "msm_accumulate_kernel<Fp>", "launches_in_trace": len(d), "avg_ms_all_launches": sum(x[1] for x in d) / len(d),
                       "timed_steps": len(timed), "avg_ms_timed_steps": sum(timed) / len(timed), "ms_each_launch": [round(x[1], 4) for x in d],
                       "hip_event_avg_ms_bench_default_run": bl["extra"]["stage_ms"]["accumulate"],
                       "_note": "from the kernel trace of the rocprofv3 --kernel-trace --stats run (--steps 20 --warmup 4, after 30 clock-priming steps and "
                                "3 one-point lane-priming launches); bench.py's roofline uses the HIP-event average of its own timed steps"}, f, indent=1)
    rows = []
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        found = glob.glob(os.path.join(src, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True)
        for path in sorted(found, key=os.path.getmtime)[-1:]:  # gpurun merges into gpurun_out/: keep the newest pass only
            acc, grid = {}, {}
            for r in csv.DictReader(open(path)):
                if r.get("Counter_Name") != ctr:
                    continue
                k = short_name(r["Kernel_Name"])
                a = acc.setdefault(k, {})
                a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
                grid.setdefault(k, {})[r["Dispatch_Id"]] = int(r.get("Grid_Size", 0) or 0)
            for k in acc:                            # launches with the kernel's largest grid only
                top = max(grid[k].values())
                acc[k] = {i: v for i, v in acc[k].items() if grid[k][i] == top}
            for k, d in sorted(acc.items()):
                avg = sum(d.values()) / len(d)
                rows.append((k, ctr, len(d), avg))
                per.setdefault(k, {})[ctr] = avg
    with open(os.path.join(dst, rnd + "_pmc_fetch_write_summary.csv"), "w") as f:
        f.write("kernel,counter,dispatches,avg_value_KB_per_dispatch\n")
        for k, ctr, n, avg in rows:
            f.write('"%s",%s,%d,%.1f\n' % (k, ctr, n, avg))
    acc_k = [k for k in per if k.startswith("msm_accumulate_kernel") and "FpTag" in k and "Fp2" not in k]
    if acc_k and len(per[acc_k[0]]) == 2:
        v = per[acc_k[0]]
        # calibrated on this kernel's own access patterns (profiles/r04_fetch_calibration.md, tools/fetch_calibration.hip): the 64-byte point
        # gathers are counted exactly (Infinity-Cache hits included), the 4-byte list words at one half -- W * n = 16 * 2^20 words per launch
        list_half = 16 * (1 << 20) * 4 / 2.0
        traffic = {"msm_accumulate_g1_2^20": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 + list_half,
                   "raw_fetch_plus_write_bytes": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0, "list_words_counted_at_one_half_bytes_added": list_half,
                   "_note": "bytes per launch = (FETCH_SIZE + WRITE_SIZE) * 1024 from two separate rocprofv3 --pmc passes (profiles/%s_pmc_fetch_write_summary.csv), "
                            "corrected as calibrated in profiles/r04_fetch_calibration.md: 64-byte gathers read exactly (no x2; the guide's 1/2 holds for "
                            "16 B/lane streaming reads), 4-byte list words read 1/2" % rnd}
        # the NTT passes of the 2^22-point transform (the bench's secondary; the profiled run transforms forward and back, both
        # directions use the same three kernels): FETCH + WRITE summed over the passes, RAW -- WRITE_SIZE is exact, the passes' loads
        # (32-byte elements as two 16-byte loads per lane in 128-byte runs, 36-byte twiddles as dwords) are not calibrated
        ntt = {k: v for k, v in per.items() if k.startswith("ntt_pass_kernel") and len(v) == 2}
        if ntt:
            traffic["ntt_2^22_per_transform_raw"] = sum(v["FETCH_SIZE"] + v["WRITE_SIZE"] for v in ntt.values()) * 1024.0
            # every pass reads its 2^22 elements of 32 bytes with 16-byte loads, which FETCH_SIZE counts at one half (pattern C of the
            # calibration; the passes behind the first read exactly 64 MB + their twiddles): add the other half of 3 x 128 MB
            traffic["ntt_2^22_per_transform_corrected"] = traffic["ntt_2^22_per_transform_raw"] + 3 * (1 << 22) * 32 / 2.0
            traffic["ntt_2^22_per_pass_raw"] = {k: {"fetch": v["FETCH_SIZE"] * 1024.0, "write": v["WRITE_SIZE"] * 1024.0} for k, v in sorted(ntt.items())}
        with open(os.path.join(dst, "traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
    fold_sq(src, dst, rnd)
    print("folded", src, "->", dst)
    return 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["run", "fold"])
    ap.add_argument("--round", default="r02")
    a = ap.parse_args()
    sys.exit(run(a.round) if a.mode == "run" else fold(a.round))
