#!/usr/bin/env python3
"""Time the channels-last BatchNorm kernels (stats / apply / backward) at the step's largest shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
L.pdgn_bn_scratch_floats.restype = ctypes.c_longlong
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for rows, C in ((179200, 1024), (358400, 512), (89600, 512), (358400, 64), (71680, 256), (358400, 16), (35840, 128), (8960, 64)):
    x = torch.randn(rows, C, device="cuda"); dy = torch.randn_like(x); mul = torch.rand_like(x)
    y = torch.empty_like(x); dx = torch.empty_like(x); dmul = torch.empty_like(x)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    stats = torch.empty(4 * C, device="cuda"); bs = torch.empty(2 * C, device="cuda")
    scr = torch.empty(L.pdgn_bn_scratch_floats(ctypes.c_longlong(rows), C), device="cuda")
    gb = rows * C * 4 / 1e9
    a = t(lambda: L.pdgn_bn_stats(ctypes.c_longlong(rows), C, ctypes.c_float(1e-5), ctypes.c_float(0.1), ptr(x), ptr(g), ptr(b), None, ptr(rm), ptr(rv), ptr(scr), ptr(stats), stream_of(x)))
    f = t(lambda: L.pdgn_bn_act_forward(ctypes.c_longlong(rows), C, 2, ptr(x), ptr(stats), None, ptr(y), 0, stream_of(x)))
    fm = t(lambda: L.pdgn_bn_act_forward(ctypes.c_longlong(rows), C, 2, ptr(x), ptr(stats), ptr(mul), ptr(y), 0, stream_of(x)))
    bw = t(lambda: L.pdgn_bn_act_backward(ctypes.c_longlong(rows), C, 2, 1, ptr(x), ptr(dy), None, ptr(stats), ptr(scr), ptr(bs), ptr(dx), None, 0, stream_of(x)))
    bm = t(lambda: L.pdgn_bn_act_backward(ctypes.c_longlong(rows), C, 2, 1, ptr(x), ptr(dy), ptr(mul), ptr(stats), ptr(scr), ptr(bs), ptr(dx), ptr(dmul), 0, stream_of(x)))
    print("rows %7d C %5d (%.0f MB): stats %6.1f us %.2f TB/s | apply %6.1f us %.2f | apply*mul %6.1f us %.2f | bwd %6.1f us %.2f | bwd*mul %6.1f us %.2f" % (
        rows, C, gb * 1e3, a, gb / a * 1e3, f, 2 * gb / f * 1e3, fm, 3 * gb / fm * 1e3, bw, 5 * gb / bw * 1e3, bm, 8 * gb / bm * 1e3))
    del x, dy, mul, y, dx, dmul
