#!/usr/bin/env python3
"""A mid-size contraction (conv2's dense half at stage 3 and its neighbours) on the 256 x 128 tile with two fp16 parts -- maxima handed
in, as the step would have them from the producing kernel -- against what the launch model picks today (three parts, its own tile)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib, fused
from pdgn_amd._lib import ptr, stream_of, check
L = _lib.lib()
_lib.set_gemm_mode("x2")


def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / it * 1e3)
    return best


def planes2(w):
    n, k = w.shape
    ld = (k + 7) // 8 * 8
    buf = torch.empty(2 * n * ld + 2 * n + 8, dtype=torch.int16, device=w.device)     # (+ the rows' maxima behind the planes)
    P = buf[:2 * n * ld].view(2, n, ld)
    check(L.pdgn_split_f16x2(n, k, ptr(w), k, ptr(P), ld, ctypes.c_longlong(n * ld), None, 0, ctypes.c_longlong(0), stream_of(w)), "split")
    return P


for (m, n, k) in [(17920, 256, 2560), (8960, 128, 1280), (17920, 6432, 64), (35840, 512, 256), (71680, 256, 256), (17920, 512, 256), (179200, 256, 64)]:
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda")
    _lib.set_gemm_config(None)
    pl3 = fused.split_planes(w, False)
    auto = t(lambda: fused.gemm_nt_planes(a, pl3.p, n, k))
    _lib.set_gemm_config(0)
    big3 = t(lambda: fused.gemm_nt_planes(a, pl3.p, n, k))
    P2 = planes2(w)
    xm = fused.operand_maxima(a)
    big2 = t(lambda: fused.gemm_nt_planes(a, P2, n, k, max_a=xm))
    _lib.set_gemm_config(None)
    print("%6d %5d %5d  auto (cfg %2d, three parts) %6.1f us | 256x128 three parts %6.1f | 256x128 two parts, maxima free %6.1f" %
          (m, n, k, L.pdgn_gemm_nt_config(ctypes.c_longlong(m), n, k, 0), auto, big3, big2))
