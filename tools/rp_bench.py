#!/usr/bin/env python3
"""The short-reduction products of an iteration (per-point tap products, stages 2-4) through gemm_nt_planes with two-part planes
and handed-in row maxima: microseconds per call and the store rate.  PDGN_RP=0: the tile kernel (gemm_x3.hip); default: the
row-panel kernel (gemm_rp.hip); PDGN_RP_STORE = 0 | 2 | 16 (plain, nt, sc1 stores), PDGN_RP_CSLABS = 1 | 2 | 4 | 8."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdgn_amd import _lib, fused  # noqa: E402

_lib.set_gemm_mode("x2")


def t(fn, it=20, rep=3):
    best = 1e9
    for _ in range(rep):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(it):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / it * 1e3)
    return best


out = []
for (m, n, k) in [(35840, 12832, 128), (17920, 6432, 64), (17920, 3232, 32), (358400, 512, 64), (179200, 256, 64)]:
    a = torch.randn(m, k, device="cuda")
    w = torch.randn(n, k, device="cuda")
    pl = fused.split_planes(w, False, rows=m, x_maxima_free=True)
    am = fused.operand_maxima(a)
    us = t(lambda: fused.gemm_nt_planes(a, pl.p, n, k, max_a=am))
    out.append("%dx%dx%d %.1f us %.2f TB/s" % (m, n, k, us, m * n * 4.0 / us / 1e6))
print("RP=%s STORE=%s CSLABS=%s PIPE=%s | " % (os.environ.get("PDGN_RP", "1"), os.environ.get("PDGN_RP_STORE", "2"), os.environ.get("PDGN_RP_CSLABS", "auto"), os.environ.get("PDGN_RP_PIPE", "1")) + " | ".join(out))
