#!/usr/bin/env python3
"""Run pdgn_gemm_tn on one shape a few times (target of rocprofv3 --pmc passes): tn_one.py M N K [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
M, N, K = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
L = _lib.lib()
dY = torch.randn(M, N, device="cuda"); X = torch.randn(M, K, device="cuda"); dW = torch.zeros(N, K, device="cuda")
for _ in range(reps):
    L.pdgn_gemm_tn(ctypes.c_longlong(M), N, K, ptr(dY), ptr(X), ptr(dW), stream_of(dY))
torch.cuda.synchronize()
