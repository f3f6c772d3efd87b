#!/usr/bin/env python3
"""Time pdgn_gemm_nt on a few shapes under the current environment (PDGN_GEMM, PDGN_NT_CFG, PDGN_NT_DBG)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import fused, _lib
shapes = [(35840, 512, 5120), (35840, 12832, 128), (71680, 1024, 256), (17920, 256, 2560)]
out = []
for (m, n, k) in shapes:
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda")
    for _ in range(3): fused.gemm_nt(a, w)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fused.gemm_nt(a, w)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    out.append("%dx%dx%d %7.1f us %6.1f TF" % (m, n, k, us, 2.0 * m * n * k / us / 1e6))
print("GEMM=%s CFG=%s DBG=%s | " % (os.environ.get("PDGN_GEMM", "x3"), os.environ.get("PDGN_NT_CFG", "-"), os.environ.get("PDGN_NT_DBG", "0")) + " | ".join(out))
