#!/usr/bin/env python3
"""Every pdgn_gemm_nt / pdgn_gemm_tn problem of one G+D step (B = 35), with its launch count, the tile configuration
pdgn_gemm_nt picks, and the time of each problem in isolation next to torch's library GEMM for the same product."""
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

NCFG = 4 if os.environ.get("PDGN_GEMM", "x3").startswith("f") else 3
tot_own = tot_lib = 0.0
extra = {}


def all_cfgs(run, key, cfg):
    """ALL_CFGS=1: the problem under every tile configuration (_lib.set_gemm_config), against the launch model's pick."""
    if os.environ.get("ALL_CFGS") != "1":
        return
    per = []
    for c_ in range(NCFG):
        _lib.set_gemm_config(c_)
        per.append(t(run, it=5))
    _lib.set_gemm_config(None)
    best = min(range(NCFG), key=lambda i: per[i])
    extra[key] = extra.get(key, "") + " | cfgs " + " ".join("%.1f" % v for v in per) + (" | best %d %s" % (
        best, "" if per[cfg & 15] <= 1.03 * per[best] else "<-- pick %d is %.0f%% slower" % (cfg & 15, 100 * (per[cfg & 15] / per[best] - 1))))


rows = []
for (kind, m, n, k), cnt in log.items():
    if kind.startswith("thin"):
        continue
    pad = lambda v: (v + 3) // 4 * 4
    if kind == "nn":
        a = torch.randn(m, pad(k), device="cuda"); w = torch.randn(pad(k), pad(n), device="cuda"); c = torch.empty(m, pad(n), device="cuda")
        run = lambda: L.pdgn_gemm_nn(ctypes.c_longlong(m), pad(n), pad(k), ptr(a), pad(k), ptr(w), pad(n), None, None, 0, ptr(c), pad(n), None, stream_of(a))
        own = t(run)
        lib = t(lambda: a.matmul(w))
        cfg = L.pdgn_gemm_nt_config(ctypes.c_longlong(m), pad(n), pad(k), 0)
        all_cfgs(run, (kind, m, n, k), cfg)
    elif kind == "nt":
        a = torch.randn(m, pad(k), device="cuda"); w = torch.randn(pad(n), pad(k), device="cuda"); c = torch.empty(m, pad(n), device="cuda")
        run = lambda: L.pdgn_gemm_nt(ctypes.c_longlong(m), pad(n), pad(k), ptr(a), pad(k), ptr(w), pad(k), None, None, 0, ptr(c), pad(n), None, stream_of(a))
        own = t(run)
        lib = t(lambda: torch.nn.functional.linear(a, w))
        cfg = L.pdgn_gemm_nt_config(ctypes.c_longlong(m), pad(n), pad(k), 0)
        all_cfgs(run, (kind, m, n, k), cfg)
    else:
        dy = torch.randn(m, pad(n), device="cuda"); x = torch.randn(m, pad(k), device="cuda"); dw = torch.zeros(pad(n), pad(k), device="cuda")
        own = t(lambda: L.pdgn_gemm_tn(ctypes.c_longlong(m), pad(n), pad(k), ptr(dy), ptr(x), ptr(dw), stream_of(dy)))
        lib = t(lambda: dy.t().matmul(x))
        cfg = -1
        if pad(n) >= 64 and pad(k) >= 64:
            runb = lambda: L.pdgn_gemm_tn_big(ctypes.c_longlong(m), pad(n), pad(k), ptr(dy), pad(n), ptr(x), pad(k), ptr(dw), 0, stream_of(dy))
            big = t(runb)
            extra[(kind, m, n, k)] = " | tn_big %8.1f us %6.1f TF (x%.2f of gemm_tn)" % (big, 2.0 * m * n * k / big / 1e6, big / own)
            all_cfgs(runb, (kind, m, n, k), L.pdgn_gemm_nt_config(ctypes.c_longlong(pad(n)), pad(k), min(m, 0x7fffffff), 0))
    rows.append((own * cnt, kind, m, n, k, cnt, cfg, own, lib))
    tot_own += own * cnt; tot_lib += lib * cnt
rows.sort(reverse=True)
for tt, kind, m, n, k, cnt, cfg, own, lib in rows:
    fl = 2.0 * m * n * k
    print("%s M%-7d N%-6d K%-6d x%-2d cfg %2d | own %8.1f us %6.1f TF | lib %8.1f us %6.1f TF | own/lib %.2f | %7.1f us/step%s" % (
        kind, m, n, k, cnt, cfg, own, fl / own / 1e6, lib, fl / lib / 1e6, own / lib, tt, extra.get((kind, m, n, k), "")))
print("total per step: own %.2f ms, library (heuristic pick) %.2f ms" % (tot_own / 1e3, tot_lib / 1e3))
