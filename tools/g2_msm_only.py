#!/usr/bin/env python3
"""Three unbound and three bound-bases G2 MSMs of 2^20 points and nothing else: the program behind profiles/r02_pmc_sq_g2.csv
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
              --output-format csv -d gpurun_out/pmc_g2 -- python3 tools/g2_msm_only.py"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
import torch
from zkhip import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]          # another build of libzkhip.so (same-session A/B traces)
from zkhip.device import MsmPlan
from zkhip.field import G2, g2_to_limbs
from zkhip.synthetic import random_scalars
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
n = 1 << 20
rng = np.random.default_rng(3)
S, K = random_scalars(rng, n), random_scalars(rng, n)
base = g2_to_limbs([G2])
P = np.zeros((n, 16), dtype=np.uint64)
_lib.check(lib.zk_fixed_base_g2(_lib.ptr(base), _lib.ptr(K), n, _lib.ptr(P)))
dS, dP = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(P.view(np.int64)).cuda()
plan = MsmPlan(_lib.GROUP_G2, n)
for _ in range(3):
    plan.run_limbs(dS.data_ptr(), dP.data_ptr(), n, st)
plan.bind(dP.data_ptr(), n, st)
for _ in range(3):
    plan.run_limbs(dS.data_ptr(), None, n, st)
torch.cuda.synchronize()
