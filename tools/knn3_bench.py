#!/usr/bin/env python3
"""pdgn_knnquery at the step's shapes: old one-query-per-wave kernel (PDGN_KNN3_WAVE4=0 in a child) vs the four-query one,
bit-exact comparison against each other, distance evaluations per second."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
B = 35
for n, m, k in [(2048, 1024, 20), (2048, 2048, 20), (1024, 512, 20), (512, 256, 20), (256, 256, 20), (2048, 512, 20), (1024, 256, 20), (2048, 256, 20), (5000, 333, 7)]:
    g = torch.Generator(device="cuda").manual_seed(n + m)
    xyz = torch.randn(B, n, 3, device="cuda", generator=g)
    q = torch.randn(B, m, 3, device="cuda", generator=g)
    idx = torch.empty(B, m, k, device="cuda", dtype=torch.int32); d2 = torch.empty(B, m, k, device="cuda")
    us = t(lambda: L.pdgn_knnquery(B, n, m, k, ptr(xyz), ptr(q), ptr(idx), ptr(d2), stream_of(xyz)))
    print("n%-5d m%-5d k%-3d %8.1f us  %6.3f T evals/s  checksum %d %.6f" % (n, m, k, us, B * n * m / us / 1e6, int(idx.long().sum()), float(d2.double().sum())), flush=True)
