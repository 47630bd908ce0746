#!/usr/bin/env python3
"""Three 2^20-point transforms one after the other against ONE batched launch per pass (zk_ntt_dev_batch): what a prover's groups of
independent transforms would gain from sharing their launches.   python3 tools/ntt_batch_probe.py [--log-n 20] [--batch 3]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--batch", type=int, default=3)
args = ap.parse_args()
import torch
from zkhip.device import NttPlan
from zkhip.synthetic import random_scalars
n, B = 1 << args.log_n, args.batch
d = torch.from_numpy(random_scalars(np.random.default_rng(1), n * B).view(np.int64)).cuda()
plan = NttPlan(args.log_n)
st = torch.cuda.current_stream().cuda_stream
def singles():
    for b in range(B):
        plan.run(d.data_ptr() + b * n * 32, False, None, st)
def batched():
    plan.run_batch(d.data_ptr(), B, False, st)
def timed(f, reps):
    for _ in range(30):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)
res = {"log_n": args.log_n, "batch": B}
for k in range(3):
    res["singles_ms_%d" % k] = timed(singles, 100)
    res["batched_ms_%d" % k] = timed(batched, 100)
print(json.dumps(res))
