#!/usr/bin/env python3
"""Library GEMM on the step's forward / input-gradient shapes under torch's two BLAS back ends."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
shapes = [(35840, 12832, 128), (35840, 512, 5120), (35840, 128, 12832), (35840, 5120, 512), (358400, 512, 64), (358400, 64, 512),
          (71680, 1024, 256), (71680, 256, 1024), (17920, 6432, 64), (17920, 256, 2560), (71680, 256, 256), (17920, 2560, 256)]
for pref in ("cublas", "cublaslt"):
    torch.backends.cuda.preferred_blas_library(pref)
    tot = 0
    for M, N, K in shapes:
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); Wt = W.t().contiguous()
        u = t(lambda: torch.nn.functional.linear(A, W))
        v = t(lambda: A.matmul(Wt))
        tot += u
        print("%-8s M%-7d N%-6d K%-6d  linear(A,W) %7.1f us %6.1f TF | A@Wt %7.1f us %6.1f TF" % (pref, M, N, K, u, 2.0 * M * N * K / u / 1e6, v, 2.0 * M * N * K / v / 1e6))
    print(pref, "sum linear us", tot)
