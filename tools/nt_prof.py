#!/usr/bin/env python3
"""A few launches of pdgn_gemm_nt per (shape, configuration) for rocprofv3 passes (kernel trace / PMC).
usage: nt_prof.py [cfg list, e.g. 0,1]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
for M, N, K in [(35840, 512, 5120), (35840, 12832, 128), (35840, 128, 12832)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    for cfg in cfgs:
        os.environ["PDGN_NT_CFG"] = str(cfg)
        for _ in range(4):
            L.pdgn_gemm_nt(ctypes.c_longlong(M), N, K, ptr(A), K, ptr(W), K, None, None, 0, ptr(C), N, None, stream_of(A))
    torch.cuda.synchronize()
    del A, W, C
