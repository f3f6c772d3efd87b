#!/usr/bin/env python3
"""pdgn_small_mlp_forward / _backward at the step's per-sample layer shapes: per-launch duration with the stream drained
between launches (these kernels sit in dependent chains: their latency, not their throughput, is what the step pays)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
R = 35
SHAPES = [(128, 4096, 1), (32, 32, 1), (64, 64, 1), (128, 128, 1), (256, 256, 1), (32, 512, 1), (64, 512, 1), (128, 512, 1),
          (256, 128, 0), (128, 64, 0), (64, 1, 0), (512, 256, 0), (256, 64, 0), (1024, 512, 0), (512, 256, 0)]


def one(fn, it=30):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(it):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for K, N, bn in SHAPES:
    x = torch.randn(R, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    g = torch.ones(N, device="cuda"); be = torch.zeros(N, device="cuda"); rm = torch.zeros(N, device="cuda"); rv = torch.ones(N, device="cuda")
    y = torch.empty(R, N, device="cuda"); pre = torch.empty(R, N, device="cuda"); st = torch.empty(2 * N, device="cuda")
    dy = torch.randn(R, N, device="cuda"); dpre = torch.empty(R, N, device="cuda"); dW = torch.empty(N, K, device="cuda")
    dg = torch.empty(N, device="cuda"); db = torch.empty(N, device="cuda"); dbias = torch.empty(N, device="cuda")
    f = one(lambda: L.pdgn_small_mlp_forward(R, K, N, 2, bn, ctypes.c_float(1e-5), ctypes.c_float(0.1), ptr(x), ptr(W), ptr(b),
                                             ptr(g) if bn else None, ptr(be) if bn else None, ptr(rm) if bn else None,
                                             ptr(rv) if bn else None, ptr(y), ptr(pre), ptr(st) if bn else None, stream_of(x)))
    bw = one(lambda: L.pdgn_small_mlp_backward(R, K, N, 2, bn, ptr(x), ptr(dy), ptr(pre), ptr(st) if bn else None,
                                               ptr(g) if bn else None, ptr(be) if bn else None, ptr(dpre), ptr(dg) if bn else None,
                                               ptr(db) if bn else None, ptr(dbias), ptr(dW), stream_of(x)))
    mm = one(lambda: dpre.matmul(W))
    emp = one(lambda: None)
    print("K=%-5d N=%-5d bn=%d  fwd %6.1f us  bwd %6.1f us  dx=dpre@W (torch) %6.1f us   (empty event pair %.1f us)" % (K, N, bn, f, bw, mm, emp))
