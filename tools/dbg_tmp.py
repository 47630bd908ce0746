import sys, os, ctypes
import numpy as np
sys.path.insert(0, "interactive-zkp-study_amd"); sys.path.insert(0,"oracle")
from zkhip import _lib
lib=_lib.load()
S=np.zeros((4,4),dtype=np.uint64); S[:,0]=5
P=np.zeros((4,8),dtype=np.uint64); P[:,0]=1; P[:,4]=2
out=np.zeros(8,dtype=np.uint64); inf=ctypes.c_int(0)
print("msm rc", lib.zk_msm_g1(_lib.ptr(S),_lib.ptr(P),4,_lib.ptr(out),ctypes.byref(inf)))
hip=ctypes.CDLL("libamdhip64.so")
n=ctypes.c_int(0)
print("hipGetDeviceCount rc", hip.hipGetDeviceCount(ctypes.byref(n)), n.value, "last", hip.hipGetLastError())
import torch
print("torch count", torch.cuda.device_count(), torch.cuda.is_available())
x=torch.zeros(4).cuda(); print("ok", x.sum().item())
