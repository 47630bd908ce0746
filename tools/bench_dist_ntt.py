#!/usr/bin/env python3
"""Timing of the four-step (multi-GPU) NTT path on however many ranks are launched (1 without torchrun):
    python tools/bench_dist_ntt.py --log-n 24
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_dist_ntt.py --log-n 24
Each rank holds its block-cyclic share (n/N elements); reports ms per forward transform and the round-trip check."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from zkhip.synthetic import random_scalars
    from zkhip.distributed import DistNtt
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    d = DistNtt(a.log_n)
    rows, cols = d.local_shape_in()
    x = torch.from_numpy(random_scalars(np.random.default_rng(3 + rank), rows * cols).view(np.int64).reshape(rows, cols, 4)).cuda()
    x0 = x.clone()
    y = d.forward(x)          # forward / inverse use their argument as scratch
    back = d.inverse(y)
    ok = bool(torch.equal(back, x0))
    x = x0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    e0.record()
    for _ in range(a.reps):
        y = d.forward(x)      # (re-transforms the scratch of the previous repetition: same work)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    if rank == 0:
        n = 1 << a.log_n
        print(json.dumps({"log_n": a.log_n, "ranks": world, "ms_per_forward": round(ms, 4), "elements_per_s": n / (ms * 1e-3), "roundtrip_exact": ok}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
