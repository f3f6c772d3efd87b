#!/usr/bin/env python3
"""The three arithmetic modes of the dense contractions against fp64 on the same operands (error relative to sum |a||w|), all three
operand layouts, over the fp32 exponent range, with the pre-split second operand; and their time on the step's largest shapes.
usage: x2_check.py [quick]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib, fused
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
dev = "cuda"


def run(mode, M, N, K, A, W, dY, rows):
    _lib.set_gemm_mode(mode)
    out = {}
    C = torch.empty(M, N, device=dev)
    assert L.pdgn_gemm_nt(ctypes.c_longlong(M), N, K, ptr(A), K, ptr(W), K, None, None, 0, ptr(C), N, None, stream_of(A)) == 0
    out["nt"] = C[:rows]
    dX = torch.empty(M, K, device=dev)
    assert L.pdgn_gemm_nn(ctypes.c_longlong(M), K, N, ptr(dY), N, ptr(W), K, None, None, 0, ptr(dX), K, None, stream_of(A)) == 0
    out["nn"] = dX[:rows]
    if N >= 64 and K >= 64:
        dW = torch.empty(N, K, device=dev)
        assert L.pdgn_gemm_tn_big(ctypes.c_longlong(rows), N, K, ptr(dY), N, ptr(A), K, ptr(dW), 0, stream_of(A)) == 0
        out["tn"] = dW
    if mode != "fp32" and N % 4 == 0 and K % 4 == 0:
        pl = fused.split_planes(W, True)
        out["ps"] = fused.gemm_nt_planes(A, pl.p, N, K)[:rows]
        out["ps_t"] = fused.gemm_nt_planes(dY, pl.t, K, N)[:rows]
    torch.cuda.synchronize()
    return out


for (M, N, K) in [(35840, 512, 5120), (35840, 12832, 128), (71680, 1024, 256), (9000, 132, 1284), (4100, 64, 6432), (20000, 64, 64), (3000, 256, 8)]:
    for scale in (1.0, 1e-20, 1e15):
        g = torch.Generator(device=dev).manual_seed(M + K)
        rows = min(M, 4096)
        A = torch.randn(M, K, device=dev, generator=g) * torch.rand(M, 1, device=dev, generator=g) * 3 * scale
        W = torch.randn(N, K, device=dev, generator=g)
        dY = torch.randn(M, N, device=dev, generator=g) * scale
        a64, w64, d64 = A[:rows].double(), W.double(), dY[:rows].double()
        ref = {"nt": a64 @ w64.t(), "nn": d64 @ w64, "tn": d64.t() @ a64}
        ref["ps"], ref["ps_t"] = ref["nt"], ref["nn"]
        mag = {"nt": a64.abs() @ w64.abs().t(), "nn": d64.abs() @ w64.abs(), "tn": d64.abs().t() @ a64.abs()}
        mag["ps"], mag["ps_t"] = mag["nt"], mag["nn"]
        line = "%6dx%5dx%4d scale %g |" % (M, N, K, scale)
        for mode in ("fp32", "x3", "x2"):
            out = run(mode, M, N, K, A, W, dY, rows)
            line += " %s:" % mode + " ".join("%s %.2e" % (k, ((o.double() - ref[k]).abs() / mag[k].clamp_min(1e-300)).max().item()) for k, o in out.items()) + " |"
        print(line, flush=True)
        if len(sys.argv) > 1:
            break
_lib.set_gemm_mode("x3")
