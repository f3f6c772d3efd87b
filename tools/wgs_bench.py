#!/usr/bin/env python3
"""wgs_fwd vs wgs_fwd_stats (+ the statistics pass it replaces) at the stage-4 inte shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib(); L.pdgn_bn_scratch_floats.restype = ctypes.c_longlong
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
b, n, k, T, P, C = 35, 1024, 10, 6, 5, 1024
ldy = 12832
Y = torch.randn(b, n, ldy, device="cuda"); idx = torch.randint(0, n, (b, n, k), device="cuda", dtype=torch.int32)
bias = torch.randn(b, C, device="cuda"); out = torch.empty(b, n, P, C, device="cuda")
rows = b * n * P
scr = torch.empty(L.pdgn_bn_scratch_floats(ctypes.c_longlong(rows), C), device="cuda")
g = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda"); st = torch.empty(4 * C, device="cuda")
a = t(lambda: L.pdgn_window_gather_sum(b, n, k, ldy, T, P, C, 0, T * C, ptr(Y), ptr(idx), ptr(bias), C, ptr(out), stream_of(Y)))
c = t(lambda: L.pdgn_window_gather_sum_stats(b, n, k, ldy, T, P, C, 0, T * C, ptr(Y), ptr(idx), ptr(bias), C, ptr(out), ptr(scr), stream_of(Y)))
d = t(lambda: L.pdgn_bn_stats(ctypes.c_longlong(rows), C, ctypes.c_float(1e-5), ctypes.c_float(0.1), ptr(out), ptr(g), ptr(be), None, ptr(rm), ptr(rv), ptr(scr), ptr(st), stream_of(Y)))
print("wgs_fwd %.1f us | wgs_fwd_stats %.1f us | bn_stats pass %.1f us  => %.1f vs %.1f" % (a, c, d, a + d, c))
# adjoint through the transposed graph (stage-4 inte shape)
rowptr = torch.empty(b, n + 1, dtype=torch.int32, device="cuda"); edges = torch.empty(b, n * k, dtype=torch.int32, device="cuda")
scr2 = torch.empty(2 * b * n, dtype=torch.int32, device="cuda")
L.pdgn_knn_graph_transpose(b, n, k, ptr(idx), ptr(rowptr), ptr(edges), ptr(scr2), stream_of(Y))
dout = torch.randn(b, n, P, C, device="cuda"); dY = torch.empty(b, n, ldy, device="cuda")
e = t(lambda: L.pdgn_window_gather_sum_backward_csr(b, n, k, ldy, T, P, C, 0, T * C, ptr(dout), ptr(rowptr), ptr(edges), ptr(dY), None, 0, stream_of(Y)))
print("wgs_bwd_csr %.1f us" % e)
# conv2's neighbour half at stage 4: T = k = 10 taps, one position, C = 2*Fout = 512
T2, P2, C2 = 10, 1, 512
ldy2 = T2 * C2 + C2
Y2 = torch.randn(b, n, ldy2, device="cuda"); out2 = torch.empty(b, n, P2, C2, device="cuda")
dout2 = torch.randn(b, n, P2, C2, device="cuda"); dY2 = torch.empty(b, n, ldy2, device="cuda")
f2 = t(lambda: L.pdgn_window_gather_sum(b, n, k, ldy2, T2, P2, C2, 0, T2 * C2, ptr(Y2), ptr(idx), ptr(bias), C, ptr(out2), stream_of(Y)))
g2 = t(lambda: L.pdgn_window_gather_sum_backward_csr(b, n, k, ldy2, T2, P2, C2, 0, T2 * C2, ptr(dout2), ptr(rowptr), ptr(edges), ptr(dY2), None, 0, stream_of(Y)))
print("conv2 half (T=10, P=1, C=512): wgs_fwd %.1f us | wgs_bwd_csr %.1f us" % (f2, g2))
# the 16-channel branches (conv_fea: T = 1, P = 10, C = 16; conv_xyz likewise): the small shapes' kernels, forward and adjoint
for (bb, nn) in ((35, 1024), (35, 512), (35, 256)):
    T3, P3, C3 = 1, 10, 16
    ldy3 = 12832 if nn == 1024 else 6432 if nn == 512 else 3232
    idx3 = torch.randint(0, nn, (bb, nn, k), device="cuda", dtype=torch.int32)
    idx3[:, :, 0] = 3                                         # a hub: every point's first neighbour
    Y3 = torch.randn(bb, nn, ldy3, device="cuda"); out3 = torch.empty(bb, nn, P3, C3, device="cuda")
    dout3 = torch.randn(bb, nn, P3, C3, device="cuda"); dY3 = torch.empty(bb, nn, ldy3, device="cuda")
    rp3 = torch.empty(bb, nn + 1, dtype=torch.int32, device="cuda"); ed3 = torch.empty(bb, nn * k, dtype=torch.int32, device="cuda")
    sc3 = torch.empty(2 * bb * nn, dtype=torch.int32, device="cuda")
    L.pdgn_knn_graph_transpose(bb, nn, k, ptr(idx3), ptr(rp3), ptr(ed3), ptr(sc3), stream_of(Y))
    f3 = t(lambda: L.pdgn_window_gather_sum(bb, nn, k, ldy3, T3, P3, C3, ldy3 - 32, ldy3 - 16, ptr(Y3), ptr(idx3), None, 0, ptr(out3), stream_of(Y)))
    g3 = t(lambda: L.pdgn_window_gather_sum_backward_csr(bb, nn, k, ldy3, T3, P3, C3, ldy3 - 32, ldy3 - 16, ptr(dout3), ptr(rp3), ptr(ed3), ptr(dY3), None, 0, stream_of(Y)))
    print("16-channel branch (T=1, P=10, C=16) n=%d: wgs_fwd %.1f us | wgs_bwd_csr %.1f us" % (nn, f3, g3))
