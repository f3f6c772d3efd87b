#!/usr/bin/env python3
"""Ablation of pdgn_gemm_nt (PDGN_NT_DBG: 1 = stores dropped) on the stage-4 shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
cfgs = [int(c) for c in os.environ.get("NT_CFGS", "0,1").split(",")]
for M, N, K in [(35840, 12832, 128), (35840, 512, 5120), (35840, 5120, 512), (71680, 256, 128)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * K
    for cfg in cfgs:
        os.environ["PDGN_NT_CFG"] = str(cfg)
        line = "M%-7d N%-6d K%-6d cfg %d" % (M, N, K, cfg)
        for dbg in [int(x) for x in os.environ.get("NT_DBGS", "0,1").split(",")]:
            os.environ["PDGN_NT_DBG"] = str(dbg)
            u = t(lambda: L.pdgn_gemm_nt(ctypes.c_longlong(M), N, K, ptr(A), K, ptr(W), K, None, None, 0, ptr(C), N, None, stream_of(A)))
            line += " | dbg%d %7.1f us %6.1f TF" % (dbg, u, fl / u / 1e6)
        os.environ["PDGN_NT_DBG"] = "0"
        print(line, flush=True)
