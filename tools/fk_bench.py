#!/usr/bin/env python3
"""pdgn_feature_knn at the step's four shapes: time (sqnorm + Gram/selection), MFMA fraction, checksum of the graph."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = ctypes.CDLL(os.environ["PDGN_FK_SO"]) if os.environ.get("PDGN_FK_SO") else _lib.lib()
_lib.lib()
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
B, k = 35, 10
for F, N in [(32, 128), (64, 256), (128, 512), (256, 1024), (256, 2048)]:
    g = torch.Generator(device="cuda").manual_seed(F + N)
    x = torch.nn.functional.leaky_relu(torch.randn(B, F, N, device="cuda", generator=g))
    idx = torch.empty(B, N, k, device="cuda", dtype=torch.int32); sq = torch.empty(B, N, device="cuda")
    us = t(lambda: L.pdgn_feature_knn(B, F, N, k, ptr(x), ptr(sq), ptr(idx), stream_of(x)))
    fl = 2.0 * B * N * N * F
    print("F%-4d N%-5d %8.1f us  %6.1f TF (%.3f of 157.3)  checksum %d" % (F, N, us, fl / us / 1e6, fl / us / 1e6 / 157.3, int(idx.long().sum())), flush=True)
