#!/usr/bin/env python3
"""A/B timing of the NTT (and a few MSMs to load the chip) with the in-tree library or with another build of it:
    python tools/ab_ntt.py new
    python tools/ab_ntt.py path/to/other/libzkhip.so
Run both in ONE gpurun call, alternately: box-to-box and clock differences are larger than most kernel changes."""
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/ -> repo root
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
import zkhip._lib as L
if len(sys.argv) > 1 and sys.argv[1] != "new":
    L.LIB_PATH = sys.argv[1]
import torch
from zkhip.device import NttPlan, MsmPlan
from zkhip.synthetic import random_scalars, arithmetic_points
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
res = {}
def ntt_time(Lg, reps):
    n = 1 << Lg
    d = torch.from_numpy(random_scalars(np.random.default_rng(1), n).view(np.int64)).cuda()
    plan = NttPlan(Lg)
    for _ in range(2):
        plan.run(d.data_ptr(), False, None, st); plan.run(d.data_ptr(), True, None, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        plan.run(d.data_ptr(), False, None, st); plan.run(d.data_ptr(), True, None, st)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / (2 * reps), 4)
res["cold_22_r5"] = ntt_time(22, 5)
res["22_r20"] = ntt_time(22, 20)
res["22_r100"] = ntt_time(22, 100)
# load the GPU with MSMs like bench.py does, then time again
n = 1 << 20
S = torch.from_numpy(random_scalars(np.random.default_rng(2), n).view(np.int64)).cuda()
P = torch.from_numpy(arithmetic_points(lib, n, 12345, 777).view(np.int64)).cuda()
plan = MsmPlan(L.GROUP_G1, n)
for _ in range(60):
    plan.run_limbs(S.data_ptr(), P.data_ptr(), n, st)
res["after_msm_22_r5"] = ntt_time(22, 5)
res["after_msm_22_r20"] = ntt_time(22, 20)
res["24_r10"] = ntt_time(24, 10)
res["20_r50"] = ntt_time(20, 50)
print(sys.argv[1] if len(sys.argv) > 1 else "new", json.dumps(res))
