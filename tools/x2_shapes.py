#!/usr/bin/env python3
"""Every dense-contraction problem of one G+D step (B = 35) alone, in the three-part bf16 mode ("x3") and in the two-part fp16 mode
("x2": its operand scans included, as the entry points run them), with the launch count per step: where the two-part mode pays."""
import ctypes, os, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib, fused
from pdgn_amd._lib import ptr, stream_of
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
L = _lib.lib()
B = int(os.environ.get("B", "35"))
tr = PDGNTrainer(device="cuda", distributed=False); tr.train()
reals = synthetic_batch(B, "cuda")
for _ in range(2):
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
fused.GEMM_LOG = []
tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
log, fused.GEMM_LOG = Counter(fused.GEMM_LOG), None
del tr


def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / it * 1e3)
    return best


pad = lambda v: (v + 3) // 4 * 4
rows = []
for (kind, m, n, k), cnt in log.items():
    if kind.startswith("thin"):
        continue
    n4, k4 = pad(n), pad(k)
    if kind == "nn":
        a = torch.randn(m, k4, device="cuda"); w = torch.randn(k4, n4, device="cuda"); c = torch.empty(m, n4, device="cuda")
        run = lambda: L.pdgn_gemm_nn(ctypes.c_longlong(m), n4, k4, ptr(a), k4, ptr(w), n4, None, None, 0, ptr(c), n4, None, stream_of(a))
    elif kind == "nt":
        a = torch.randn(m, k4, device="cuda"); w = torch.randn(n4, k4, device="cuda"); c = torch.empty(m, n4, device="cuda")
        run = lambda: L.pdgn_gemm_nt(ctypes.c_longlong(m), n4, k4, ptr(a), k4, ptr(w), k4, None, None, 0, ptr(c), n4, None, stream_of(a))
    else:
        if not (n4 >= 64 and k4 >= 64):
            continue
        dy = torch.randn(m, n4, device="cuda"); x = torch.randn(m, k4, device="cuda"); dw = torch.zeros(n4, k4, device="cuda")
        run = lambda: L.pdgn_gemm_tn_big(ctypes.c_longlong(m), n4, k4, ptr(dy), n4, ptr(x), k4, ptr(dw), 0, stream_of(dy))
    res = {}
    for mode in ("x3", "x2"):
        _lib.set_gemm_mode(mode)
        res[mode] = t(run)
    cfg = L.pdgn_gemm_nt_config(ctypes.c_longlong(m), n4, k4, 0) if kind != "tn" else -1
    rows.append((cnt * (res["x3"] - res["x2"]), kind, m, n, k, cnt, cfg, res["x3"], res["x2"]))
rows.sort(reverse=True)
print("kind      m      n      k  count cfg    x3 us    x2 us   gflop   saved us/step")
tot3 = tot2 = 0.0
for save, kind, m, n, k, cnt, cfg, u3, u2 in rows:
    tot3 += cnt * u3; tot2 += cnt * u2
    print("%-3s %7d %6d %6d %5d %4d %8.1f %8.1f %7.2f %10.1f" % (kind, m, n, k, cnt, cfg, u3, u2, 2.0 * m * n * k / 1e9, save))
print("sum per step: x3 %.1f us, x2 %.1f us" % (tot3, tot2))
