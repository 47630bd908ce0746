#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM throughput on MI355X (BASELINE.json metric: "BN254 G1 MSM points/sec").

One "step" = one full multi-scalar multiplication over 2^20 synthetic random scalars/points that
are already resident in HBM (BASELINE.json configs[1]); with --gpus N every rank holds its own
2^20-point chunk (weak scaling), computes its partial sum with the same HIP pipeline, and the
partials are combined by one RCCL all-gather + host fold per step (zkhip.distributed).

    python bench.py                       # 1 GPU
    python bench.py --gpus N              # N GPUs: starts N ranks itself (torch.distributed.run children, before this
                                          #   process touches a GPU), relays rank 0's line, exits with the children's code;
                                          #   fails if fewer than N devices are visible (--rehearse: ranks share the
                                          #   visible devices over gloo -- a correctness rehearsal, not a measurement)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W      # the same ranks, launched by the caller

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      dominant kernel (bucket accumulation): algorithmic bytes (96 B/point) / its average
                launch duration measured with HIP events on the pipeline's stream, vs 8 TB/s HBM.
  cpu_baseline  the reference-shaped pure-Python path (oracle/py_ref.py: per-term affine
                double-and-add + affine add, as zkp/plonk/kzg.py:59-65 does) timed on one host
                core on a bounded sample of the same workload.
"""
import argparse, ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))

HBM_PEAK_GBS = 8000.0
G1_BYTES_PER_POINT = 96  # 32 B scalar + 64 B affine point, each read once


def arithmetic_source_hash():
    """SHA-256 over the sources that determine what one bucket addition compiles to: csrc/field.h, csrc/curve.h and the accumulate
    kernel's text in csrc/msm_impl.h (between its [accumulate-kernel-begin] / -end markers).  Figures that were NOT measured by this run but read off
    a compiled code object or a committed counter pass (profiles/static_counts.json, profiles/*_pmc_sq_summary.csv) carry the hash of
    the sources they were taken from and are dropped from the line when it no longer matches."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "interactive-zkp-study_amd", "csrc")
    for name in ("field.h", "curve.h"):
        h.update(open(os.path.join(csrc, name), "rb").read())
    text = open(os.path.join(csrc, "msm_impl.h")).read()
    a, b = text.find("// [accumulate-kernel-begin]"), text.find("// [accumulate-kernel-end]")
    h.update((text[a:b] if 0 <= a < b else text).encode())     # markers gone: hash the whole file (nothing static is quoted then)
    return h.hexdigest()


def static_counts():
    """profiles/static_counts.json when it belongs to the current arithmetic sources, else None."""
    path = os.path.join(ROOT, "profiles", "static_counts.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    return rec if rec.get("arithmetic_source_sha256") == arithmetic_source_hash() else None


from zkhip.synthetic import (R_MOD, arithmetic_dot_device, arithmetic_points, limbs_dot_mod_r, random_scalars,  # noqa: E402
                             random_scalars_device)


def oracle_g1_mul(k):
    """k * G1 as plain integers (x, y) | None from the C oracle -- the CHECKER of the closed forms below (the expected point of
    every `verified_closed_form`), never the thing measured: the timed results come from libzkhip alone."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import py_ref
    return c_oracle.g1_mul(py_ref.G1, int(k) % R_MOD)


def oracle_g2_mul(k):
    """k * G2 as ((x0, x1), (y0, y1)) | None from the C oracle (checker only, like oracle_g1_mul)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import py_ref
    return c_oracle.g2_mul(py_ref.G2, int(k) % R_MOD)


def g2_point_ints(pt):
    return None if pt is None else tuple(tuple(int(c) for c in v.coeffs) for v in pt)


def blocking_ms(plan, d_s, d_p, n, stream, reps=10):
    """One MSM on the GPU at a time: submit, wait for the result, submit the next -- what a single commitment or proof element costs a
    caller that needs the point before it can go on (a Fiat-Shamir round); min and mean over `reps` calls after two untimed ones."""
    import torch
    for _ in range(2):
        res = plan.run(d_s, d_p, n, stream)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = plan.run(d_s, d_p, n, stream)
        ts.append((time.perf_counter() - t0) * 1e3)
    return res, round(min(ts), 4), round(sum(ts) / len(ts), 4)


def point_ints(pt):
    """A facade G1 point (FQ, FQ) | None as plain integers, the oracle's format."""
    return None if pt is None else (int(pt[0]), int(pt[1]))


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def committed_issue_rate(kernel):
    """The kernel's vector-ALU issue rate from the committed counter pass (profiles/*_pmc_sq_summary.csv, written by
    tools/collect_profiles.py: SQ_INSTS_VALU per SIMD cycle, the kernel running alone under rocprofv3 --pmc).  Not measured by this
    run: quoted next to the live numbers, with its source; None when the file is absent."""
    import csv, glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_sq_summary.csv")))
    if not found:
        return None
    meta = found[-1][:-4] + ".meta.json"
    try:
        tagged = json.load(open(meta)).get("arithmetic_source_sha256")
    except (OSError, ValueError):
        tagged = None
    if tagged != arithmetic_source_hash():
        return None                                 # taken from other arithmetic sources, or not tagged at all: do not quote it
    for r in csv.DictReader(open(found[-1])):
        if r["kernel"] == kernel:
            v = float(r["valu_insts_per_simd_cycle"])
            return {"source": os.path.relpath(found[-1], ROOT), "insts_per_simd_cycle": v, "peak": 0.25, "frac": v / 0.25,
                    "arithmetic_source_sha256": tagged,
                    "note": "a wave64 vector instruction occupies a SIMD for 4 cycles; counters of a separate rocprofv3 --pmc run, not of this one"}
    return None


def launch_command(argv, n_ranks, port):
    """The child command of `bench.py --gpus N`: one rank per GPU under torch.distributed.run, same arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(args, argv):
    """`bench.py --gpus N` with no rank environment: start the N ranks as children of this process, which has not made
    (and never makes) a HIP call -- torch.cuda.device_count() only reads the device list -- and relay their output."""
    import torch
    visible = torch.cuda.device_count()
    if args.dry_launch:
        print(json.dumps({"launch": launch_command(argv, args.gpus, 0), "visible_devices": visible}), flush=True)
        return 0
    if visible < args.gpus and not args.rehearse:
        sys.stderr.write("bench.py: --gpus %d needs %d HIP devices, %d visible (use --rehearse to let the ranks share them over gloo)\n"
                         % (args.gpus, args.gpus, visible))
        return 2
    if visible < 1:
        sys.stderr.write("bench.py needs a HIP device (there is no CPU fallback)\n")
        return 2
    cmd = launch_command(argv, args.gpus, _free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def host_description():
    """CPU model and core count of the box the CPU baselines run on (BASELINE.md section 3 asks for both)."""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count()}


def fit_and_extrapolate(sizes, secs, targets, nlogn):
    """Least-squares fit of t = a * f(n) through the timed sizes (f = n, or n log2 n for the recursive fft) -> a, the relative
    residuals and the extrapolated seconds at the target sizes.  SURVEY.md section 8 row D4: 'fitted (linear in n for MSM, n log n for
    NTT) and extrapolated to the config sizes ... clearly labelled'."""
    f = lambda n: n * np.log2(n) if nlogn else float(n)
    x = np.array([f(n) for n in sizes], dtype=np.float64)
    y = np.array(secs, dtype=np.float64)
    a = float((x * y).sum() / (x * x).sum())
    return {"model": "t = a * n * log2(n)" if nlogn else "t = a * n", "a_seconds": a,
            "measured": [{"n": int(n), "seconds": round(float(t), 4), "fit_seconds": round(a * f(n), 4)} for n, t in zip(sizes, secs)],
            "max_rel_residual": float(np.max(np.abs(a * x - y) / y)),
            "extrapolated_seconds": {"2^%d" % L: a * f(1 << L) for L in targets},
            "label": "reference-shaped pure Python on ONE host core, EXTRAPOLATED from the measured sizes (not measured at these sizes)"}


def cpu_baseline_msm(args, plan, scalars, points, d_scalars, d_points, n, stream, result):
    """The `cpu_baseline` object of the line (row D4): oracle/py_ref.msm_naive -- per-term affine double-and-add + affine add, the
    reference's zkp/plonk/kzg.py:59-65 -- on one host core at n = 2^6 .. 2^10 plus --cpu-sample points (each equal to the GPU MSM of
    the same sample), a linear fit extrapolated to 2^20 / 2^24 / 2^26, and the oracle's C Pippenger over the host threads on the
    full workload.  `result`: the timed GPU result to compare the C line with (None: a fresh GPU run of the same chunk)."""
    from zkhip import _lib
    from zkhip.field import limbs_to_g1
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import py_ref
    import c_oracle
    k = min(args.cpu_sample, n)
    sc = _lib.limbs_to_ints(scalars[:k])
    pts = [(int(p[0]), int(p[1])) for p in limbs_to_g1(points[:k])]

    def timed(m):
        c0 = time.perf_counter()
        ref_pt = py_ref.msm_naive(sc[:m], pts[:m])
        dt = time.perf_counter() - c0
        sub = plan.run(d_scalars.data_ptr(), d_points.data_ptr(), m, stream)
        return dt, (sub is None and ref_pt is None) or (sub is not None and ref_pt == (int(sub[0]), int(sub[1])))

    fit_sizes = [m for m in (64, 128, 256, 512, 1024) if m <= k]
    fit_t, fit_ok = [], True
    for m in fit_sizes:
        dt, ok = timed(m)
        fit_t.append(dt)
        fit_ok = fit_ok and ok
    cdt, same = timed(k)
    fit = fit_and_extrapolate(fit_sizes + [k], fit_t + [cdt], (20, 24, 26), False) if fit_sizes else None
    if fit:
        fit["every_sample_equals_gpu"] = bool(fit_ok and same)
    # second CPU line ("strong CPU"): the oracle's bucket-method MSM in C with its windows spread over the host threads
    threads = max(1, min(16, os.cpu_count() or 1))
    c1 = time.perf_counter()
    c_pt = c_oracle.g1_msm_bucket_mt_arr(scalars, points, 16, threads)
    cct = time.perf_counter() - c1
    if result is None:
        result = plan.run_limbs(d_scalars.data_ptr(), d_points.data_ptr(), n, stream)
    same_c = bool((not result[1]) and np.array_equal(result[0], c_pt))
    compiled = {"value": n / cct, "unit": "points/s", "cores": threads, "kind": "port",
                "sample": "all %d points of rank 0's workload, oracle/bn254_oracle.c orc_g1_msm_bucket_mt (Pippenger, c=16, Jacobian, windows over %d "
                          "threads); %.1f s; bit-identical to the GPU result for the same points: %s" % (n, threads, cct, same_c)}
    return {"value": k / cdt, "unit": "points/s", "cores": 1, "kind": "port",
            "sample": "first %d points/scalars of the same workload, oracle/py_ref.msm_naive "
                      "(affine double-and-add per term, as zkp/plonk/kzg.py:59-65); %.1f s; matches GPU MSM of the same sample: %s"
                      % (k, cdt, same),
            "host": host_description(), "extrapolation": fit, "compiled_c": compiled}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log-n", type=int, default=20, help="log2 of points per GPU")
    ap.add_argument("--ntt-log-n", type=int, default=22, help="log2 size of the secondary NTT measurement (0 = skip)")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="points timed on the pure-Python baseline (0 = skip)")
    ap.add_argument("--groth16-log-m", type=int, default=20, help="log2 constraints of the secondary Groth16 prove() timing (0 = skip)")
    ap.add_argument("--plonk-log-n", type=int, default=20, help="log2 gates of the secondary device-resident PLONK prove() timing (0 = skip)")
    ap.add_argument("--no-bound", action="store_true", help="skip the secondary bound-bases run (keeps a profiler's per-kernel averages to the headline workload)")
    ap.add_argument("--no-facade", action="store_true", help="skip the secondary toy-size facade proofs (their small MSMs would enter a profiler's per-kernel averages)")
    ap.add_argument("--no-g2", action="store_true", help="skip the secondary G2 measurements")
    ap.add_argument("--no-witness-like", action="store_true", help="skip the secondary skewed-scalar run (keeps a profiler's per-kernel averages to the headline workload)")
    ap.add_argument("--shard-total-log", type=int, default=26, help="N > 1 only: log2 points of the secondary ONE-MSM-sharded-over-all-GPUs "
                    "measurement (BASELINE.json configs[4]; 0 = skip)")
    ap.add_argument("--dist-groth16-log-m", type=int, default=20, help="N > 1 only (power-of-two N): log2 constraints of the Groth16 proof with every vector "
                    "distributed over the ranks (0 = skip)")
    ap.add_argument("--dist-ntt-log-n", type=int, default=24, help="N > 1 only: log2 size of the single NTT spread over all GPUs (0 = skip)")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-GPU code path (process group, all-gather, fold) even with one rank")
    ap.add_argument("--no-rewarm", action="store_true", help="N > 1: no untimed steps between the opening barrier and the start of the clock")
    ap.add_argument("--stagger-us", type=int, default=0, help="pause between the first submissions of a run of steps, so that the lanes "
                    "start a third of a step apart instead of together")
    ap.add_argument("--rehearse", action="store_true", help="N > 1 with fewer than N devices: the ranks share the visible devices and talk over gloo "
                    "(checks the N-rank code path; the numbers are not a scaling measurement)")
    ap.add_argument("--dry-launch", action="store_true", help="N > 1: print the child command instead of running it")
    ap.add_argument("--exchange", choices=["rccl", "host"], default="rccl", help="N > 1: how the 128-byte MSM partials travel: one RCCL all-gather per "
                    "step on a side stream (default; what north_star names), or a host-side gloo all-gather of the host-resident partials "
                    "(no GPU work at all; for comparison)")
    ap.add_argument("--sizes", default="24,26", help="N = 1: log2 sizes of the extra one-GPU G1 MSM measurements (BASELINE.json metric names 2^20/2^24/2^26; '' = skip)")
    ap.add_argument("--sizes-ntt", default="24", help="N = 1: log2 sizes of the extra NTT measurements ('' = skip)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    if args.gpus > 1 or args.force_dist:
        # The multi-rank path adds the exchange's side stream and RCCL's own streams to the three lane streams: give the HIP
        # runtime eight hardware queues instead of its default four, so that no two of them have to share one (read when the
        # runtime initialises, i.e. before the first HIP call below; 1.45 -> 1.43 ms per step on one rank over RCCL).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import torch
    import torch.distributed as dist
    from zkhip import _lib
    from zkhip.device import MsmPlan, NttPlan
    from zkhip.distributed import ExchangeWorker, sharded_msm
    from zkhip.field import limbs_to_g1

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    visible = torch.cuda.device_count()
    if visible < world and not args.rehearse:
        raise SystemExit("bench.py: %d ranks but %d HIP devices visible (one process per GPU; --rehearse shares devices over gloo)" % (world, visible))
    rehearsal = args.rehearse and visible < world
    dev_index = local_rank % visible
    torch.cuda.set_device(dev_index)
    lib = _lib.load()
    _lib.check(lib.zk_set_device(dev_index))
    dev = torch.device("cuda", dev_index)
    dist_on = world > 1 or args.force_dist
    n = 1 << args.log_n
    # The MSM plan first: HIP maps streams to its few hardware queues in creation order, and two lanes that end up on one
    # queue run their kernels one after the other (measured: 1.50 instead of 1.38 ms per step when the process group's
    # streams were created first).  The lanes take the first queues; RCCL's streams then share with whatever is left.
    plan = MsmPlan(_lib.GROUP_G1, n)
    plan.set_profiling(True)
    # collectives: RCCL ("nccl") between GPUs; gloo when ranks share a device (RCCL refuses two ranks on one GPU), in which
    # case the few bytes of every exchange are staged through the host (zkhip.distributed handles both)
    cdev = None if rehearsal else dev                                  # where control tensors and partials live
    if dist_on:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        world = dist.get_world_size()                                  # n_gpus of the line = what the process group reports
        # which physical devices the ranks sit on: every rank contributes its device's UUID (PCI bus id where the runtime has none),
        # so that a scaling line proves N distinct GPUs (or shows a rehearsal's shared one)
        props = torch.cuda.get_device_properties(dev_index)
        my_dev = str(getattr(props, "uuid", None) or "pci:%s:%s:%s" % (getattr(props, "pci_domain_id", "?"), getattr(props, "pci_bus_id", "?"),
                                                                          getattr(props, "pci_device_id", "?")))
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "host": socket.gethostname(), "device": my_dev, "local_index": dev_index})
        host_group = dist.new_group(backend="gloo") if (args.exchange == "host" and not rehearsal) else None

    # ---- synthetic workload, generated once and left resident in HBM
    rng = np.random.default_rng(0x5EEDB254 + rank)
    scalars = random_scalars(rng, n)
    ks = random_scalars(rng, n)  # P_i = k_i * G1 -> closed form (sum s_i k_i) * G1
    g1 = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
    points = np.zeros((n, 8), dtype=np.uint64)
    _lib.check(lib.zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(ks), n, _lib.ptr(points)))
    d_scalars = torch.from_numpy(scalars.view(np.int64)).to(dev)
    d_points = torch.from_numpy(points.view(np.int64)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    # Steps are pipelined (zk_msm_submit / zk_msm_collect, plan.max_in_flight() = 3 outstanding): every
    # submission runs in its own workspace and stream, so consecutive MSMs overlap on the GPU and the
    # ~0.2 ms host fold of step k hides behind the following steps, as in a prover issuing its MSMs back to
    # back.  Every step's pipeline, read-back, fold (and for N > 1 its all-gather) completes inside the
    # timed region.
    depth = plan.max_in_flight() if n <= (1 << 22) else 1   # larger MSMs already run as 2^22-point chunks through all lanes
    xdev, xgroup = (None, host_group) if (dist_on and host_group is not None) else (cdev, None)

    # the exchange side of the multi-rank step loop runs on a thread of its own (zkhip.distributed.ExchangeWorker)
    worker = ExchangeWorker(_lib.GROUP_G1, device=xdev, group=xgroup, cuda_device=dev_index) if dist_on else None

    def run_steps(k, stage_acc=None):
        """k complete steps: every step's pipeline, read-back, host fold and -- for N > 1 -- its exchange and rank-order fold
        finish before this returns.  A lane is refilled as soon as its result has been collected; for N > 1 the collected
        partial goes to the exchange thread (ExchangeWorker), which is flushed before this returns."""
        res, pending, submitted = None, [], 0

        def submit():
            nonlocal submitted
            pending.append(plan.submit(d_scalars.data_ptr(), d_points.data_ptr(), n, stream))
            submitted += 1

        def collect():
            out = plan.collect_partial(pending.pop(0)) if dist_on else plan.collect_limbs(pending.pop(0))
            if stage_acc is not None:
                np.add(stage_acc, plan.stage_ms(), out=stage_acc)
            return out

        while submitted < k and len(pending) < depth:
            if pending and args.stagger_us:      # see --stagger-us
                t_go = time.perf_counter() + args.stagger_us * 1e-6
                while time.perf_counter() < t_go:
                    pass
            submit()
        trace = [] if os.environ.get("ZK_BENCH_TRACE") else None   # diagnostics: host time of every collect, to stderr
        tr0 = time.perf_counter()
        while pending:
            out = collect()
            if trace is not None:
                trace.append(round((time.perf_counter() - tr0) * 1e3, 2))
            if submitted < k:
                submit()
            if dist_on:
                worker.post(out)
            else:
                res = out
        if dist_on:
            res = worker.flush()
        if trace is not None:
            sys.stderr.write("TRACE k=%d end=%.2f collects at %s\n" % (k, (time.perf_counter() - tr0) * 1e3, trace))
        return res

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed priming before the W warm-up steps: after the idle setup phase the GPU needs some tens of ms of load to
    # reach its sustained clocks; with a small W the first timed steps would otherwise run on a cold clock.
    run_steps(30)
    if args.warmup:
        run_steps(args.warmup)
    stage = np.zeros(4)
    fence()
    if dist_on and not args.no_rewarm:
        # The barrier costs the host a millisecond or two, in which the chip idles and clocks down (the first timed steps
        # then cost 2 ms more on one rank than without the barrier: profiles/r02_experiments.md).  Three more untimed steps
        # bring the clock back; every rank does the same three steps, so the ranks still start together, and the device is
        # synchronised once more before the clock starts.
        run_steps(3)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    result = run_steps(args.steps, stage)
    t_steps = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    t_fence = elapsed - t_steps
    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stage /= max(args.steps, 1)

    # ---- correctness of the timed result: closed form (sum_i s_i k_i mod r) * G1
    local_dot = limbs_dot_mod_r(scalars, ks)
    if dist_on:
        dots = [None] * world
        dist.all_gather_object(dots, local_dot)
        total_dot = sum(dots) % R_MOD
        got = result
    else:
        total_dot = local_dot
        got = None if result[1] else limbs_to_g1(result[0])[0]
    verified = (point_ints(got) == oracle_g1_mul(total_dot))           # expected point from the oracle, not from the library

    extra = {"verified_closed_form": bool(verified), "window_bits": plan.window_bits(n),
             "timed_region_ms": {"steps": round(t_steps * 1e3, 3), "closing_barrier_and_sync": round(t_fence * 1e3, 3),
                                 "total_max_over_ranks": round(elapsed * 1e3, 3)},
             "stage_ms": {"prepare": round(float(stage[0]), 4), "sort": round(float(stage[1]), 4), "accumulate": round(float(stage[2]), 4),
                          "reduce": round(float(stage[3]), 4)}}

    # ---- secondary, N > 1: ONE MSM of 2^26 points sharded over the ranks by contiguous chunks (BASELINE.json configs[4]).
    # Every rank joins in (collectives inside); any failure is reported on every rank alike before the collectives start.
    if dist_on and args.shard_total_log:
        try:
            from zkhip.distributed import shard_range
            n_tot = 1 << args.shard_total_log
            lo, hi = shard_range(n_tot, rank, world)
            m_loc = hi - lo
            d_s2 = random_scalars_device(m_loc, dev, 0x5EEDB260 + rank)
            p_loc = arithmetic_points(lib, m_loc, first=lo)                  # P_i = (k0 + i d) G for the global index i
            d_p2 = torch.from_numpy(p_loc.view(np.int64)).to(dev)
            del p_loc
            plan2 = MsmPlan(_lib.GROUP_G1, m_loc)
            ok_all = torch.ones(1, device=cdev)
        except Exception as exc:                                              # noqa: BLE001 -- keep the ranks in step
            ok_all = torch.zeros(1, device=cdev)
            sys.stderr.write("sharded_one_msm set-up failed on rank %d: %r\n" % (rank, exc))
        dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
        if float(ok_all.item()) == 1.0:
            one = lambda: sharded_msm(_lib.GROUP_G1, plan2.run_partial(d_s2.data_ptr(), d_p2.data_ptr(), m_loc, stream), device=cdev)
            one()
            fence()
            t_s = time.perf_counter()
            sreps = 3
            for _ in range(sreps):
                got_tot = one()
            fence()
            sms = (time.perf_counter() - t_s) / sreps * 1e3
            dots = [None] * world
            dist.all_gather_object(dots, arithmetic_dot_device(d_s2, first=lo))
            extra["sharded_one_msm"] = {"log_n_total": args.shard_total_log, "points_per_gpu": m_loc, "ms_per_msm": round(sms, 3),
                                        "points_per_s": n_tot / (sms * 1e-3), "verified_closed_form": bool(point_ints(got_tot) == oracle_g1_mul(sum(dots) % R_MOD))}
            plan2.close()
            del d_s2, d_p2
        else:
            extra["sharded_one_msm"] = {"error": "setup failed on some rank"}

    # ---- secondary: the same MSM with the bases bound to the plan (zk_msm_plan_bind_points: what a prover does with its
    # CRS / SRS); the one-off table build is outside the timed loop, as a prover's key loading is
    if rank == 0 and world == 1 and n > (1 << 17) and not args.no_bound:
        tb0 = time.perf_counter()
        plan.bind(d_points.data_ptr(), n, stream)
        bind_s = time.perf_counter() - tb0
        pend, bres = [], None
        for _ in range(10):
            pend.append(plan.submit(d_scalars.data_ptr(), None, n, stream))
            if len(pend) == depth:
                bres = plan.collect_limbs(pend.pop(0))
        for t in pend:
            bres = plan.collect_limbs(t)
        pend = []
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        bsteps = 60
        for _ in range(bsteps):
            pend.append(plan.submit(d_scalars.data_ptr(), None, n, stream))
            if len(pend) == depth:
                bres = plan.collect_limbs(pend.pop(0))
        for t in pend:
            bres = plan.collect_limbs(t)
        bms = (time.perf_counter() - tb0) / bsteps * 1e3
        extra["bound_bases"] = {"ms_per_step": round(bms, 4), "points_per_s": n / (bms * 1e-3), "bind_ms": round(bind_s * 1e3, 1),
                                "table_bytes": 13 * n * 64, "same_result_as_unbound": bool(np.array_equal(bres[0], result[0]) and bres[1] == result[1])}
        # the same two modes as BLOCKING calls (one MSM on the chip at a time): the figure a prover's serial rounds see
        _, b_min, b_mean = blocking_ms(plan, d_scalars.data_ptr(), None, n, stream)
        plan.bind(None, 0, stream)
        ures, u_min, u_mean = blocking_ms(plan, d_scalars.data_ptr(), d_points.data_ptr(), n, stream)
        extra["blocking"] = {"what": "uniform 2^%d-point G1 MSM, one call at a time (submit -> result), min / mean of 10" % args.log_n,
                             "unbound_ms_per_msm_blocking": u_min, "unbound_mean_ms": u_mean,
                             "bound_ms_per_msm_blocking": b_min, "bound_mean_ms": b_mean,
                             "verified_closed_form": bool(point_ints(ures) == oracle_g1_mul(local_dot))}

    # ---- secondary: G2 (proof_b's query, zkp/groth16/proving.py:35-45): 2^20 points P_i = k_i * G2, blocking calls --
    # uniform scalars with unbound and bound bases, and witness-like scalars (a quarter 0, a quarter 1) -- each against the oracle's
    # closed form (sum s_i k_i) * G2
    if rank == 0 and world == 1 and n > (1 << 17) and not args.no_g2:
        try:
            from zkhip.field import G2 as G2_GEN, g2_to_limbs
            p2 = np.zeros((n, 16), dtype=np.uint64)
            _lib.check(lib.zk_fixed_base_g2(_lib.ptr(g2_to_limbs([G2_GEN])), _lib.ptr(ks), n, _lib.ptr(p2)))
            d_p2 = torch.from_numpy(p2.view(np.int64)).to(dev)
            del p2
            plan2 = MsmPlan(_lib.GROUP_G2, n)
            plan2.set_profiling(True)
            want = oracle_g2_mul(local_dot)
            g2res, g_min, g_mean = blocking_ms(plan2, d_scalars.data_ptr(), d_p2.data_ptr(), n, stream, 6)
            g2 = {"unbound": {"ms_per_msm_blocking": g_min, "mean_ms": g_mean, "stage_ms": [round(v, 4) for v in plan2.stage_ms()],
                              "verified_closed_form": bool(g2_point_ints(g2res) == want)}}
            wl2 = scalars.copy()
            pick2 = np.random.default_rng(0x5EEDB257).random(n)
            wl2[pick2 < 0.25] = 0
            wl2[(pick2 >= 0.25) & (pick2 < 0.5)] = np.array([1, 0, 0, 0], dtype=np.uint64)
            d_wl2 = torch.from_numpy(wl2.view(np.int64)).to(dev)
            want_wl = oracle_g2_mul(limbs_dot_mod_r(wl2, ks))
            wres, w_min, w_mean = blocking_ms(plan2, d_wl2.data_ptr(), d_p2.data_ptr(), n, stream, 6)
            g2["witness_like"] = {"ms_per_msm_blocking": w_min, "mean_ms": w_mean, "stage_ms": [round(v, 4) for v in plan2.stage_ms()],
                                  "ratio_to_uniform": round(w_min / g_min, 3), "verified_closed_form": bool(g2_point_ints(wres) == want_wl)}
            tb0 = time.perf_counter()
            plan2.bind(d_p2.data_ptr(), n, stream)
            bind2 = (time.perf_counter() - tb0) * 1e3
            bres, gb_min, gb_mean = blocking_ms(plan2, d_scalars.data_ptr(), None, n, stream, 6)
            g2["bound"] = {"ms_per_msm_blocking": gb_min, "mean_ms": gb_mean, "stage_ms": [round(v, 4) for v in plan2.stage_ms()],
                           "bind_ms": round(bind2, 1), "verified_closed_form": bool(g2_point_ints(bres) == want)}
            wbres, wb_min, wb_mean = blocking_ms(plan2, d_wl2.data_ptr(), None, n, stream, 6)
            g2["bound_witness_like"] = {"ms_per_msm_blocking": wb_min, "mean_ms": wb_mean, "verified_closed_form": bool(g2_point_ints(wbres) == want_wl)}
            extra["g2"] = g2
            plan2.close()
            del d_p2, d_wl2
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001 -- a secondary measurement must not take the headline line down with it
            extra["g2"] = {"error": repr(exc)}

    # ---- secondary: NTT forward + inverse round trip (BASELINE.json configs[2]).  With N > 1 every rank transforms a
    # polynomial of its own (batch-parallel mode: no exchange) between barriers; the aggregate is N transforms per span.
    if args.ntt_log_n:
        L = args.ntt_log_n
        m = 1 << L
        coeffs = random_scalars(np.random.default_rng(0x5EEDB255 + rank), m)
        d = torch.from_numpy(coeffs.view(np.int64)).to(dev)
        ref = d.clone()
        nplan = NttPlan(L)
        # The chip clocks down while it idles and takes a few milliseconds of work to come back: five timed round trips straight
        # after a synchronisation read 0.63 ms per 2^22-point transform, twenty after ten untimed ones 0.56, a hundred 0.51 (round 2;
        # round 5: 0.57 / 0.51 / 0.46-0.47, tools/ab_ntt.py).  Sixty untimed round trips (~60 ms of work) first, like the headline's
        # priming steps, then forty timed: the sustained rate, which is what a prover's transforms run at behind its MSMs.
        ntt_warm = 60
        for _ in range(ntt_warm):
            nplan.run(d.data_ptr(), False, None, stream)
            nplan.run(d.data_ptr(), True, None, stream)
        reps = 40
        fence()
        tn0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            nplan.run(d.data_ptr(), False, None, stream)
            nplan.run(d.data_ptr(), True, None, stream)
        e1.record()
        fence()
        span = time.perf_counter() - tn0
        ms = e0.elapsed_time(e1) / (2 * reps)  # per transform, this rank's device time
        exact = bool(torch.equal(d, ref))
        if dist_on:
            tt = torch.tensor([span, 0.0 if exact else 1.0], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            span, exact = float(tt[0].item()), float(tt[1].item()) == 0.0
        extra["ntt"] = {"log_n": L, "ms_per_transform": round(ms, 4), "elements_per_s": m / (ms * 1e-3),
                        "algorithmic_GBps": 64.0 * m / (ms * 1e-3) / 1e9, "hbm_frac": 64.0 * m / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "roundtrip_exact": exact, "timed_round_trips": reps, "warmup_round_trips": ntt_warm}
        # the same in the shape of the headline's roofline object (algorithmic bytes: 64 B per element and transform);
        # traffic: the committed counter passes' FETCH + WRITE over the three pass kernels (16-byte loads corrected x2 as calibrated:
        # profiles/r04_fetch_calibration.md), for this size only
        ntt_traffic = None
        try:
            ntt_traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("ntt_2^%d_per_transform_corrected" % L)
        except Exception:  # noqa: BLE001
            pass
        extra["ntt"]["roofline"] = {"bound": "hbm", "kernel": "ntt_pass_kernel (all passes of one transform)", "achieved": extra["ntt"]["algorithmic_GBps"],
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": extra["ntt"]["hbm_frac"], "traffic": ntt_traffic,
                                    "note": "issue-bound (0.20-0.21 of 0.25 vector instructions per SIMD cycle: profiles/*_pmc_sq_summary.csv); ~160 instructions per element and stage"}
        if dist_on:
            agg = world * 2 * reps * m / span   # wall clock between barriers (includes launch latency), all ranks
            extra["ntt"]["all_gpus"] = {"mode": "one polynomial per GPU, no exchange", "elements_per_s": agg,
                                        "hbm_frac_per_gpu": 64.0 * agg / world / 1e9 / HBM_PEAK_GBS}
        if rank == 0 and world == 1 and args.cpu_sample:
            # (one rank only: with N > 1 the other ranks would sit in their next collective through a pure-Python fft)
            try:
                # CPU lines beside the NTT (row D4): the reference-shaped recursive fft (polynomial.py:292-341) in pure Python on prefixes
                # of 2^8 .. 2^14 and 2^16 coefficients (n log n fit, extrapolated to the config sizes), and the C oracle's iterative NTT on
                # the whole vector -- each compared with the GPU output
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import py_ref
                import c_oracle
                fit_n, fit_t, all_same = [], [], True
                for Ls in [v for v in (8, 10, 12, 14, 16) if v <= L] or [L]:
                    ms_ = 1 << Ls
                    small = torch.from_numpy(coeffs[:ms_].copy().view(np.int64)).to(dev)
                    NttPlan(Ls).run(small.data_ptr(), False, None, stream)
                    torch.cuda.synchronize()
                    ints = _lib.limbs_to_ints(coeffs[:ms_])
                    w_s = py_ref.get_root_of_unity(ms_)
                    p0 = time.perf_counter()
                    py_out = py_ref.fft(ints, w_s)
                    pdt = time.perf_counter() - p0
                    same_py = _lib.limbs_to_ints(small.cpu().numpy().view(np.uint64).reshape(-1, 4)) == [int(v) for v in py_out]
                    all_same = all_same and same_py
                    fit_n.append(ms_)
                    fit_t.append(pdt)
                nfit = fit_and_extrapolate(fit_n, fit_t, (22, 24), True)
                nfit["every_sample_equals_gpu"] = bool(all_same)
                nplan.run(d.data_ptr(), False, None, stream)
                torch.cuda.synchronize()
                gpu_fwd = d.cpu().numpy().view(np.uint64).reshape(-1, 4)
                nplan.run(d.data_ptr(), True, None, stream)
                c0_ = time.perf_counter()
                c_out = c_oracle.ntt_arr(coeffs, py_ref.get_root_of_unity(m))
                cdt_ = time.perf_counter() - c0_
                extra["ntt"]["cpu_baseline"] = {
                    "value": ms_ / pdt, "unit": "elements/s", "cores": 1, "kind": "port",
                    "sample": "first 2^%d coefficients, oracle/py_ref.fft (recursive radix-2, as zkp/plonk/polynomial.py:292-341); %.2f s; "
                              "equals the GPU transform of the same sample: %s" % (Ls, pdt, same_py),
                    "host": host_description(), "extrapolation": nfit,
                    "compiled_c": {"value": m / cdt_, "unit": "elements/s", "cores": 1, "kind": "port",
                                   "sample": "all 2^%d coefficients, oracle/bn254_oracle.c orc_ntt; %.2f s; bit-identical to the GPU forward transform: %s"
                                             % (L, cdt_, bool(np.array_equal(gpu_fwd, c_out)))}}
            except Exception as exc:  # noqa: BLE001 -- a secondary CPU line must not take the headline line down with it
                extra["ntt"]["cpu_baseline"] = {"error": repr(exc)}
        del d, ref
        # ONE transform of 2^24 points spread over the ranks (four-step, one all-to-all; zkhip.distributed.DistNtt)
        if dist_on and args.dist_ntt_log_n:
            # set-up may fail on one rank only (memory): agree first, so that no rank waits alone in the all-to-all
            dn = x0 = None
            try:
                from zkhip.distributed import DistNtt
                dn = DistNtt(args.dist_ntt_log_n)
                rows, cols = dn.local_shape_in()
                x0 = torch.from_numpy(random_scalars(np.random.default_rng(0x5EEDB270 + rank), rows * cols).view(np.int64).reshape(rows, cols, 4)).to(dev)
            except Exception as exc:  # noqa: BLE001 -- a secondary measurement must not take the headline line down with it
                dn = None
                sys.stderr.write("dist_ntt set-up failed on rank %d: %r\n" % (rank, exc))
            ready = torch.tensor([1.0 if dn is not None else 0.0], dtype=torch.float64, device=cdev)
            dist.all_reduce(ready, op=dist.ReduceOp.MIN)
            if float(ready.item()) == 1.0:
                x = x0.clone()
                back = dn.inverse(dn.forward(x))
                ok_rt = bool(torch.equal(back, x0))
                x = x0.clone()
                fence()
                td0 = time.perf_counter()
                dreps = 3
                for _ in range(dreps):
                    y = dn.forward(x)
                    x = y if dn.l1 == dn.l2 else x0.clone()     # even log n: the output layout is the input layout again
                fence()
                dms = (time.perf_counter() - td0) / dreps * 1e3
                tt = torch.tensor([dms, 0.0 if ok_rt else 1.0], dtype=torch.float64, device=cdev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                extra["dist_ntt"] = {"log_n": args.dist_ntt_log_n, "ms_per_forward": round(float(tt[0].item()), 4),
                                     "elements_per_s": (1 << args.dist_ntt_log_n) / (float(tt[0].item()) * 1e-3), "roundtrip_exact": float(tt[1].item()) == 0.0}
                del x, back, y
            else:
                extra["dist_ntt"] = {"error": "setup failed on some rank"}
            del x0, dn

    # ---- secondary, N > 1 (or --force-dist): Groth16 prove() with every vector distributed over the ranks (zkhip.groth16.prover_dist:
    # the rank's constraint rows, one all-to-all per transform, the rank's slice of every query, one all-gather of the partial sums)
    if dist_on and args.dist_groth16_log_m and (world & (world - 1)) == 0:
        dg = None
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_groth16
            from zkhip.groth16.prover_dist import DistScaleCRS, DistScaleProver
            from zkhip.groth16.prover_ntt import ChainCircuit
            circ = ChainCircuit(args.dist_groth16_log_m, seed=7)
            wit = circ.witness()[0]
            toxic = dict(alpha=3926, beta=3604, gamma=2971, delta=1357, x=3721 + (1 << 201))
            tg0 = time.perf_counter()
            dcrs = DistScaleCRS(circ, toxic["alpha"], toxic["beta"], toxic["gamma"], toxic["delta"], toxic["x"])
            t_dsetup = time.perf_counter() - tg0
            dg = DistScaleProver(dcrs, device=cdev)
            d_wit = torch.from_numpy(_lib.ints_to_limbs(wit).view(np.int64)).to(dev)
        except Exception as exc:  # noqa: BLE001 -- agree first: no rank may wait alone in a collective
            dg = None
            sys.stderr.write("dist_groth16 set-up failed on rank %d: %r\n" % (rank, exc))
        ready = torch.tensor([1.0 if dg is not None else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(ready, op=dist.ReduceOp.MIN)
        if float(ready.item()) == 1.0:
            # One proof first, then every rank says whether its own went through BEFORE anyone starts the timed ones: a failure that
            # is local to a rank and shows after its collectives (a device error surfaced at collect_partial, say) is then an error
            # entry on every rank instead of a rank missing from the next all-to-all.  (A rank that dies INSIDE a collective leaves
            # the others to the process group's timeout; nothing short of a second process group can turn that into an entry.)
            proof, first_ok = None, 1.0
            try:
                proof = dg.prove(d_wit, 4106, 4565)
            except Exception as exc:  # noqa: BLE001
                first_ok = 0.0
                sys.stderr.write("dist_groth16 proof failed on rank %d: %r\n" % (rank, exc))
            agreed = torch.tensor([first_ok], dtype=torch.float64, device=cdev)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            try:
                if float(agreed.item()) != 1.0:
                    raise RuntimeError("a rank's first distributed proof failed (see stderr)")
                fence()
                tms = []
                for _ in range(4):
                    tg0 = time.perf_counter()
                    proof = dg.prove(d_wit, 4106, 4565)
                    fence()
                    tms.append((time.perf_counter() - tg0) * 1e3)
                tt = torch.tensor([min(tms)], dtype=torch.float64, device=cdev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                if rank == 0:
                    extra["dist_groth16"] = {"log_m": args.dist_groth16_log_m, "prove_ms": round(float(tt.item()), 3), "prove_ms_all_rank0": [round(v, 3) for v in tms],
                                             "setup_s_per_rank": round(t_dsetup, 3), "coefficients_per_rank": dcrs.cn, "g1_bases_per_rank": dcrs.n_g1,
                                             "verified_closed_form": bool(bench_groth16.proof_equals_oracle(circ, toxic, wit, 4106, 4565, proof))}
            except Exception as exc:  # noqa: BLE001
                extra["dist_groth16"] = {"error": repr(exc)}
            del dg, dcrs, d_wit
            torch.cuda.empty_cache()
        else:
            extra["dist_groth16"] = {"error": "setup failed on some rank"}

    # ---- secondary: "witness-like" scalars: half the scalars are 0 or 1, the rest uniform
    if rank == 0 and world == 1 and not args.no_witness_like:
        wrng = np.random.default_rng(0x5EEDB256)
        wl = scalars.copy()
        pick = wrng.random(n)
        wl[pick < 0.25] = 0
        wl[(pick >= 0.25) & (pick < 0.5)] = np.array([1, 0, 0, 0], dtype=np.uint64)
        d_wl = torch.from_numpy(wl.view(np.int64)).to(dev)
        plan.run(d_wl.data_ptr(), d_points.data_ptr(), n, stream)
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        wreps = 5
        for _ in range(wreps):
            wres = plan.run(d_wl.data_ptr(), d_points.data_ptr(), n, stream)
        wms = (time.perf_counter() - w0) / wreps * 1e3
        extra["witness_like"] = {"ms_per_msm_blocking": round(wms, 4), "points_per_s": n / (wms * 1e-3),
                                 "verified_closed_form": bool(point_ints(wres) == oracle_g1_mul(limbs_dot_mod_r(wl, ks))),
                                 "stage_ms": [round(v, 4) for v in plan.stage_ms()]}
        del d_wl

    # ---- secondary, N = 1: the other sizes BASELINE.json's metric names (2^24, 2^26 points on ONE GPU; blocking calls, closed-form
    # verified) and the 2^24-point NTT north_star names.  Inputs are generated outside the timed calls and resident in HBM.
    if rank == 0 and world == 1 and not dist_on and (args.sizes or args.sizes_ntt):
        sizes = {"msm_g1": [], "ntt": []}
        for L in [int(v) for v in args.sizes.split(",") if v]:
            try:
                nn = 1 << L
                d_s = random_scalars_device(nn, dev, 0x5EEDB300 + L)
                d_p = torch.from_numpy(arithmetic_points(lib, nn).view(np.int64)).to(dev)       # P_i = (k0 + i d) G1
                pl = MsmPlan(_lib.GROUP_G1, nn)
                pl.set_profiling(True)
                res = pl.run(d_s.data_ptr(), d_p.data_ptr(), nn, stream)
                ts = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    res = pl.run(d_s.data_ptr(), d_p.data_ptr(), nn, stream)
                    ts.append(time.perf_counter() - t0_)
                ms_ = min(ts) * 1e3
                acc_ms_ = pl.stage_ms()[2]
                chunk = min(nn, 1 << 22)                                      # larger MSMs run as 2^22-point chunks through the lanes
                gbs = G1_BYTES_PER_POINT * chunk / (acc_ms_ * 1e-3) / 1e9 if acc_ms_ > 0 else 0.0
                sizes["msm_g1"].append({"log_n": L, "ms_per_msm_blocking": round(ms_, 3), "points_per_s": nn / (ms_ * 1e-3),
                                        "accumulate_ms_last_chunk": round(acc_ms_, 4), "chunk_points": chunk, "chunks": nn // chunk,
                                        "accumulate_GBps": gbs, "accumulate_hbm_frac": gbs / HBM_PEAK_GBS,
                                        "verified_closed_form": bool(point_ints(res) == oracle_g1_mul(arithmetic_dot_device(d_s)))})
                pl.close()
                del d_s, d_p, pl
                torch.cuda.empty_cache()
            except Exception as exc:  # noqa: BLE001 -- a secondary measurement must not take the headline line down with it
                sizes["msm_g1"].append({"log_n": L, "error": repr(exc)})
        for L in [int(v) for v in args.sizes_ntt.split(",") if v]:
            try:
                mm = 1 << L
                d = random_scalars_device(mm, dev, 0x5EEDB340 + L)
                ref = d.clone()
                npl = NttPlan(L)
                for _ in range(3):
                    npl.run(d.data_ptr(), False, None, stream)
                    npl.run(d.data_ptr(), True, None, stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                nreps = 8
                e0.record()
                for _ in range(nreps):
                    npl.run(d.data_ptr(), False, None, stream)
                    npl.run(d.data_ptr(), True, None, stream)
                e1.record()
                torch.cuda.synchronize()
                ms_ = e0.elapsed_time(e1) / (2 * nreps)
                sizes["ntt"].append({"log_n": L, "ms_per_transform": round(ms_, 4), "elements_per_s": mm / (ms_ * 1e-3),
                                     "algorithmic_GBps": 64.0 * mm / (ms_ * 1e-3) / 1e9, "hbm_frac": 64.0 * mm / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "roundtrip_exact": bool(torch.equal(d, ref))})
                npl.close()
                del d, ref, npl
                torch.cuda.empty_cache()
            except Exception as exc:  # noqa: BLE001
                sizes["ntt"].append({"log_n": L, "error": repr(exc)})
        extra["sizes"] = sizes

    # ---- the integer-ALU ceiling the MSM is really priced against (row D3): this library's own field
    # multiplication / mixed addition run flat out on the whole chip, measured live
    alu = None
    if rank == 0:
        rate = ctypes.c_double()
        _lib.check(lib.zk_measure_rate(0, ctypes.byref(rate)))
        modmul_peak = rate.value
        _lib.check(lib.zk_measure_rate(1, ctypes.byref(rate)))
        madd_peak = rate.value
        _lib.check(lib.zk_measure_rate(2, ctypes.byref(rate)))
        mad_rate = rate.value
        W = -(-255 // plan.window_bits(n))
        madds = float(n) * W  # one mixed addition per (point, window) digit; zero digits (2^-c of them) are skipped
        acc_s = float(stage[2]) * 1e-3
        alu = {"unit": "G1 mixed additions/s", "per_launch": madds, "achieved": madds / acc_s if acc_s > 0 else 0.0, "peak": madd_peak,
               "frac": (madds / acc_s / madd_peak) if acc_s > 0 else 0.0, "modmul_peak_per_s": modmul_peak,
               "modmul_per_madd": 10, "note": "peak = zk_measure_rate(1): dependent XYZZ+=affine chains, one wave per workgroup, chip oversubscribed",
               "v_mad_u64_u32_lane_ops_per_s": mad_rate}
        # the ceiling that does not depend on this library's arithmetic: the chip's v_mad_u64_u32 issue rate (zk_measure_rate(2): bare
        # independent multiply-adds, as tools/ubench.hip) against the multiply-adds one mixed addition compiles to -- a figure read off
        # the compiled code object, quoted only while profiles/static_counts.json belongs to the current sources
        sc = static_counts()
        if sc:
            mpm = sc["mads_per_madd"]
            alu["mad_floor"] = {"mads_per_madd": mpm, "floor_ms": madds * mpm / mad_rate * 1e3, "frac": (madds * mpm / mad_rate) / acc_s if acc_s > 0 else 0.0,
                                "source": "profiles/static_counts.json", "arithmetic_source_sha256": sc["arithmetic_source_sha256"]}
        pmc = committed_issue_rate("msm_accumulate_kernel<zk::Fe<zk::FpTag> >")
        if pmc:
            alu["valu_issue_pmc"] = pmc

    # ---- secondary: Groth16 prove() wall-clock on a synthetic 2^20-constraint R1CS (BASELINE.json configs[3])
    if args.groth16_log_m and world == 1:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_groth16
            extra["groth16_prove"] = bench_groth16.run(args.groth16_log_m, 6)
        except Exception as exc:  # the headline number must not depend on the secondary measurement
            extra["groth16_prove"] = {"error": repr(exc)}
        # the same prover on a witness of the kind real circuits have: half of the wires are bits (they enter the merged proof_C query
        # directly and fill a few buckets with hundreds of thousands of points: the heavy-bucket path)
        try:
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            extra["groth16_prove_bool_witness"] = bench_groth16.run(args.groth16_log_m, 4, circuit="bool")
        except Exception as exc:  # noqa: BLE001
            extra["groth16_prove_bool_witness"] = {"error": repr(exc)}

    # ---- secondary: the reference-signature facade at the reference's own sizes (BASELINE.json configs[0]: toy Groth16 / PLONK proofs
    # through the host-buffer ABI, one library call per primitive; the entry points keep their plans)
    if rank == 0 and world == 1 and not dist_on and not args.no_facade:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_facade
            extra["facade_toy_ms"] = bench_facade.run(5)
        except Exception as exc:  # noqa: BLE001
            extra["facade_toy_ms"] = {"error": repr(exc)}

    # ---- secondary: PLONK prove() on a synthetic 2^20-gate circuit, all vectors resident in HBM
    if args.plonk_log_n and world == 1:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_plonk
            extra["plonk_prove"] = bench_plonk.run(args.plonk_log_n, 3)
        except Exception as exc:
            extra["plonk_prove"] = {"error": repr(exc)}

    # ---- CPU baseline: reference-shaped pure-Python path on a bounded sample, on rank 0's host cores (for N > 1 the sample and the
    # C line are rank 0's chunk of the workload; the other ranks wait at the closing barrier)
    cpu = None
    if args.cpu_sample and rank == 0:
        cpu = cpu_baseline_msm(args, plan, scalars, points, d_scalars, d_points, n, stream, None if dist_on else result)
    if rank == 0:
        total_points = n * world * args.steps
        acc_ms = float(stage[2])
        achieved = G1_BYTES_PER_POINT * n / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("msm_accumulate_g1_2^%d" % args.log_n)
            except Exception:
                traffic = None
        line = {
            "metric": "bn254_g1_msm_points_per_sec", "value": total_points / elapsed, "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BN254 G1 MSM, 2^%d uniform random scalars x random points per GPU, inputs resident in HBM "
                                   "(BASELINE.json configs[1])" % args.log_n,
                       "points_per_gpu": n, "sharding": "point chunks, 1 all-gather of 128-B partials" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate_kernel<Fp>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "integer-ALU-bound kernel: ~160 modular multiplications per point; see 'alu' and DESIGN.md", "alu": alu},
            "cpu_baseline": cpu,
            "extra": extra,
        }
        if dist_on:
            line["config"]["ranks_seen"] = {"ranks": len(seen), "distinct_devices": len({(r["host"], r["device"]) for r in seen}), "by_rank": seen}
            line["config"]["collectives"] = ("gloo, ranks share %d device(s): REHEARSAL, not a scaling measurement" % visible if rehearsal else
                                             "RCCL (nccl backend)" + ("; MSM partials over a host-side gloo group (--exchange host)" if host_group is not None else ""))
        print(json.dumps(line), flush=True)
    if dist_on:
        worker.close()
        dist.barrier()
        dist.destroy_process_group()
    if not verified:
        raise SystemExit("bench.py: MSM result does not match the closed form")


if __name__ == "__main__":
    main()
