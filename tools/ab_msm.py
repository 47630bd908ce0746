#!/usr/bin/env python3
"""A/B timing of the pipelined 2^20-point G1 MSM step (bench.py's headline loop) with the in-tree library or another build of it:
    python tools/ab_msm.py new [steps]
    python tools/ab_msm.py path/to/other/libzkhip.so [steps]
Run both in ONE gpurun call, alternately (new, old, new, old): box-to-box and clock differences exceed most kernel changes.
Prints one JSON line: ms per step over `steps` pipelined submissions (three in flight), the blocking time, per-stage HIP-event times,
and whether the result equals the closed form."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
import zkhip._lib as L
if len(sys.argv) > 1 and sys.argv[1] != "new":
    L.LIB_PATH = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
import torch
from zkhip.device import MsmPlan
from oracle_check import msm_result_is
from zkhip.synthetic import arithmetic_dot_device, arithmetic_points, random_scalars_device
lib = L.load()
n = 1 << 20
dev = torch.device("cuda", 0)
S = random_scalars_device(n, dev, 0x5EEDB300)
P = torch.from_numpy(arithmetic_points(lib, n).view(np.int64)).to(dev)
st = torch.cuda.current_stream().cuda_stream
plan = MsmPlan(L.GROUP_G1, n)
plan.set_profiling(True)


def run(k):
    pend, res, stage = [], None, np.zeros(4)
    for i in range(k):
        if len(pend) == plan.max_in_flight():
            res = plan.collect_limbs(pend.pop(0))
            stage += plan.stage_ms()
        pend.append(plan.submit(S.data_ptr(), P.data_ptr(), n, st))
    for t in pend:
        res = plan.collect_limbs(t)
        stage += plan.stage_ms()
    return res, stage / k


run(40)
torch.cuda.synchronize()
out = {"lib": L.LIB_PATH if len(sys.argv) > 1 and sys.argv[1] != "new" else "in-tree", "steps": steps}
ts = []
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, stage = run(steps)
    ts.append((time.perf_counter() - t0) / steps * 1e3)
out["ms_per_step"] = [round(t, 4) for t in ts]
out["stage_ms_pipelined"] = [round(float(v), 4) for v in stage]
tb = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r1 = plan.run(S.data_ptr(), P.data_ptr(), n, st)
    tb.append((time.perf_counter() - t0) * 1e3)
out["blocking_ms"] = round(min(tb), 4)
out["stage_ms_blocking"] = [round(float(v), 4) for v in plan.stage_ms()]
out["verified"] = bool(msm_result_is(r1, arithmetic_dot_device(S)))   # expectation from the C oracle (tools/oracle_check.py)
print(json.dumps(out), flush=True)
