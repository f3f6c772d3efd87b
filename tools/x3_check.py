#!/usr/bin/env python3
"""pdgn_gemm_nt / nn / tn_big in the running mode (PDGN_GEMM=x2 (default) | x3 | fp32) against fp64: error relative to sum_k |a| |w|, and time."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import fused, _lib

torch.manual_seed(0)
dev = "cuda"


def err(c, a64, w64):
    ref = a64 @ w64.t()
    scale = a64.abs() @ w64.abs().t()
    return ((c.double() - ref).abs() / scale.clamp_min(1e-30)).max().item()


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


if len(sys.argv) > 1:
    _lib.set_gemm_shape(int(sys.argv[1]))
print("mode", _lib.gemm_mode(), "cfg", os.environ.get("PDGN_NT_CFG", "auto"), "shape", _lib.set_gemm_shape(0))
for (m, n, k) in [(1000, 64, 36), (4099, 132, 100), (35840, 512, 5120), (35840, 12832, 128), (71680, 1024, 256), (358400, 512, 64),
                  (17920, 256, 2560), (35840, 256, 128)]:
    a = torch.randn(m, k, device=dev) * torch.rand(m, 1, device=dev) * 3
    w = torch.randn(n, k, device=dev)
    b = torch.randn(n, device=dev)
    small = m * n <= 40e6
    c = fused.gemm_nt(a, w)
    cfg = _lib.lib().pdgn_gemm_nt_config(ctypes.c_longlong(m), n, k, 0)
    msg = "nt M%-7d N%-6d K%-5d cfg %2d" % (m, n, k, cfg)
    if small:
        msg += "  err %.2e" % err(c, a.double(), w.double())
    else:
        rows = torch.randint(0, m, (2048,), device=dev)
        msg += "  err %.2e" % err(c[rows], a[rows].double(), w.double())
    us = timeit(lambda: fused.gemm_nt(a, w))
    msg += "  %8.1f us %6.1f TF" % (us, 2.0 * m * n * k / us / 1e6)
    # with bias + stats
    y, part = fused.gemm_nt(a, w, b, want_stats=True)
    if small:
        ref = (a.double() @ w.double().t() + b.double())
        msg += "  bias+stats err %.2e" % ((y.double() - ref).abs().max() / ref.abs().max()).item()
    us = timeit(lambda: fused.gemm_nt(a, w, b, want_stats=True))
    msg += "  stats %8.1f us" % us
    # nn: dx = dy @ W (W: n x k -> result m x k)
    dy = torch.randn(m, n, device=dev)
    dx = fused.gemm_nt(dy, w, w_transposed=True)
    rows = torch.randint(0, m, (1024,), device=dev)
    msg += " | nn err %.2e" % err(dx[rows], dy[rows].double(), w.double().t().contiguous())
    us = timeit(lambda: fused.gemm_nt(dy, w, w_transposed=True))
    msg += " %8.1f us %6.1f TF" % (us, 2.0 * m * n * k / us / 1e6)
    if n >= 64 and k >= 128 or n >= 128 and k >= 64:
        dw = torch.empty(n, k, device=dev)
        L = _lib.lib()
        def tn():
            _lib.check(L.pdgn_gemm_tn_big(ctypes.c_longlong(m), n, k, _lib.ptr(dy), dy.stride(0), _lib.ptr(a), a.stride(0), _lib.ptr(dw), 0,
                                          _lib.stream_of(dy)), "tn_big")
        tn()
        msg += " | tn err %.2e" % err(dw, dy.double().t().contiguous(), a.double().t().contiguous())
        us = timeit(tn)
        msg += " %8.1f us %6.1f TF" % (us, 2.0 * m * n * k / us / 1e6)
    print(msg, flush=True)
    del a, w, c, dy, dx
