#!/usr/bin/env python3
"""pdgn_gemm_tn on the weight-gradient shapes of one G+D step: logs the (M, N, K) of every call in a real
step, then times each distinct shape in isolation (HIP events) and prints TFLOP/s and the share of the step."""
import collections, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch

L = _lib.lib()
calls = collections.Counter()


class Proxy:
    def __getattr__(self, name):
        f = getattr(L, name)
        if name != "pdgn_gemm_tn":
            return f

        def wrapped(m, n, k, *rest):
            calls[(m.value, n, k)] += 1
            return f(m, n, k, *rest)
        return wrapped


B = int(sys.argv[1]) if len(sys.argv) > 1 else 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
_lib._lib = Proxy()
tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
_lib._lib = L
del tr
torch.cuda.empty_cache()


def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


tot = 0.0
rows = []
for (M, N, K), cnt in calls.items():
    mk = (lambda *sh: torch.zeros(*sh, device="cuda")) if os.environ.get("TN_ZERO") == "1" else (lambda *sh: torch.randn(*sh, device="cuda"))
    dY = mk(M, N); X = mk(M, K); dW = torch.zeros(N, K, device="cuda")
    f = lambda: L.pdgn_gemm_tn(ctypes.c_longlong(M), N, K, ptr(dY), ptr(X), ptr(dW), stream_of(dY))
    dW.zero_(); f(); ref = dY.t().matmul(X)
    err = ((dW - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    us = t(f)
    rows.append((us * cnt, M, N, K, cnt, us, 2.0 * M * N * K / us / 1e6, err))
    tot += us * cnt
    del dY, X, dW, ref
for r in sorted(rows, reverse=True):
    print("M%-7d N%-6d K%-5d x%-2d %8.1f us %6.1f TF  (%.3f ms/step)  relerr %.1e" % (r[1], r[2], r[3], r[4], r[5], r[6], r[0] / 1e3, r[7]))
print("total gemm_tn per step: %.3f ms" % (tot / 1e3))
