#!/usr/bin/env python3
"""pdgn_gemm_nt vs the library GEMM, alone and next to an HBM-bound copy stream / a small-kernel stream on other HIP
streams (what the step's side streams do to the default stream's GEMMs)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
side = torch.cuda.Stream()
big = torch.randn(64 * 1024 * 1024, device="cuda"); big2 = torch.empty_like(big)
small = torch.randn(35 * 256, 64, device="cuda"); sw = torch.randn(64, 64, device="cuda")

def measure(fn, load):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    stop = False
    if load == "copy":
        with torch.cuda.stream(side):
            for _ in range(40): big2.copy_(big)
    elif load == "small":
        with torch.cuda.stream(side):
            for _ in range(3000): torch.nn.functional.linear(small, sw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); e.synchronize()
    dt = s.elapsed_time(e) / 10 * 1e3
    torch.cuda.synchronize()
    return dt

for M, N, K in [(35840, 512, 5120), (35840, 12832, 128), (71680, 1024, 256), (35840, 256, 512)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    own = lambda: L.pdgn_gemm_nt(ctypes.c_longlong(M), N, K, ptr(A), K, ptr(W), K, None, None, 0, ptr(C), N, None, stream_of(A))
    lib = lambda: torch.nn.functional.linear(A, W)
    line = "M%-6d N%-6d K%-5d" % (M, N, K)
    for name, fn in (("own", own), ("lib", lib)):
        t0, t1, t2 = measure(fn, None), measure(fn, "copy"), measure(fn, "small")
        line += " | %s alone %7.1f  +copy %7.1f (x%.2f)  +small %7.1f (x%.2f)" % (name, t0, t1, t1 / t0, t2, t2 / t0)
    print(line, flush=True)
