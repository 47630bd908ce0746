import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
import numpy as np, torch
from bench import random_scalars, limbs_dot_mod_r
from zkhip import _lib
from zkhip.device import MsmPlan
from zkhip.field import G1, ec_mul, limbs_to_g1
lib = _lib.load()
n = 1 << 20
rng = np.random.default_rng(5)
S, K = random_scalars(rng, n), random_scalars(rng, n)
g1 = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
P = np.zeros((n, 8), dtype=np.uint64)
_lib.check(lib.zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(K), n, _lib.ptr(P)))
dS, dP = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(P.view(np.int64)).cuda()
want = ec_mul(G1, limbs_dot_mod_r(S, K))
side = torch.cuda.Stream()
plans = {}
def measure(label, st, depth, steps=40):
    plan = plans.setdefault("p", MsmPlan(_lib.GROUP_G1, n))
    def loop(k):
        pend, res = [], None
        for i in range(k):
            pend.append(plan.submit(dS.data_ptr(), dP.data_ptr(), n, st))
            if len(pend) == depth:
                res = plan.collect_limbs(pend.pop(0))
        while pend:
            res = plan.collect_limbs(pend.pop(0))
        return res
    loop(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = loop(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print("%-28s ms/step %.4f ok %s" % (label, dt, limbs_to_g1(res[0])[0] == want), flush=True)
for rep in range(4):
    measure("null stream depth 3", 0, 3)
    measure("side stream depth 3", side.cuda_stream, 3)
    measure("null stream depth 2", 0, 2)
    measure("null stream depth 1", 0, 1)
