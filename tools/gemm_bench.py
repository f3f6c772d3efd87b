#!/usr/bin/env python3
"""pdgn_gemm_nt vs torch (rocBLAS) on the step's forward / input-gradient shapes."""
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
shapes = [(35840, 12832, 128), (35840, 12832, 256), (35840, 512, 5120), (35840, 128, 12832), (35840, 5120, 512), (358400, 512, 64), (358400, 64, 512),
          (71680, 1024, 256), (71680, 256, 1024), (71680, 256, 128), (71680, 128, 64), (17920, 6432, 64), (17920, 256, 2560), (71680, 256, 256), (358400, 64, 16)]
if len(sys.argv) > 1 and sys.argv[1] == "small":           # the latency-bound NT shapes of stages 1-3 and of the discriminators
    shapes = [(4480, 1600, 32), (4480, 64, 640), (8960, 256, 32), (8960, 128, 64), (8960, 256, 128), (8960, 3232, 32), (89600, 64, 16),
              (89600, 128, 64), (8960, 128, 1280), (17920, 256, 64), (17920, 128, 64), (17920, 256, 128), (17920, 512, 256), (17920, 6432, 64),
              (179200, 64, 16), (179200, 256, 64), (35840, 256, 128), (35840, 128, 64), (35840, 512, 256), (71680, 128, 64), (71680, 256, 128),
              (71680, 256, 256), (8960, 64, 256), (17920, 64, 256), (35840, 64, 256), (71680, 64, 256)]
for M, N, K in shapes:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    st = torch.empty((M + 127) // 128 * 2 * N, device="cuda")
    f1 = lambda: L.pdgn_gemm_nt(ctypes.c_longlong(M), N, K, ptr(A), ptr(W), None, None, ptr(C), None, stream_of(A))
    f2 = lambda: L.pdgn_gemm_nt(ctypes.c_longlong(M), N, K, ptr(A), ptr(W), None, None, ptr(C), ptr(st), stream_of(A))
    f3 = lambda: torch.nn.functional.linear(A, W)
    ref = torch.nn.functional.linear(A, W)
    f1(); err = ((C - ref).abs().max() / ref.abs().max()).item()
    fl = 2.0 * M * N * K
    u1, u2, u3 = t(f1), t(f2), t(f3)
    print("M%-7d N%-6d K%-6d  mine %7.1f us %6.1f TF | +stats %7.1f us | rocblas %7.1f us %6.1f TF | relerr %.1e" % (M, N, K, u1, fl / u1 / 1e6, u2, u3, fl / u3 / 1e6, err))
