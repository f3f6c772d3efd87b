#!/usr/bin/env python3
"""Time the fused BatchNorm+LeakyReLU+slot-softmax forward and the softmax adjoint at the stage-4 size."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
M, k, C = 35840, 10, 512
x = torch.randn(M * k, C, device="cuda")
stats = torch.cat([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).cuda()
w = torch.empty(M, k // 2, 2 * C, device="cuda")
dw = torch.randn_like(w); dh = torch.empty_like(x)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
gb = x.numel() * 4 / 1e9
u = t(lambda: L.pdgn_bn_softmax_slots_permute(ctypes.c_longlong(M), k, C, 2, ptr(x), ptr(stats), ptr(w), stream_of(x)))
print("bn_softmax_perm fwd  %.1f us  %.2f TB/s" % (u, 2 * gb / u * 1e3))
u = t(lambda: L.pdgn_softmax_slots_permute(ctypes.c_longlong(M), k, C, ptr(x), ptr(w), stream_of(x)))
print("softmax_perm fwd     %.1f us  %.2f TB/s" % (u, 2 * gb / u * 1e3))
u = t(lambda: L.pdgn_softmax_slots_permute_backward(ctypes.c_longlong(M), k, C, ptr(w), ptr(dw), ptr(dh), stream_of(x)))
print("softmax_perm bwd     %.1f us  %.2f TB/s" % (u, 3 * gb / u * 1e3))
