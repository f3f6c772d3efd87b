#!/usr/bin/env python3
"""The dynamic range of the operands the two-part contractions see in one G+D iteration (B = 35): per operand log2(max / rms), the
share of its elements and of its energy (sum of squares) more than 2^-16 below the largest magnitude of THEIR SCALING UNIT -- where
the two-part split (csrc/gemm_x3.hip NP = 2) stops keeping a value's own 22 bits.  Round 6 scales every row of an operand as the
kernel sees it (rows of a row-major operand, columns of a transposed one: `unit` below); the last two columns are what the same
operand lost under round 5's one scale per operand."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import fused
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch

B = int(os.environ.get("B", "35"))
tr = PDGNTrainer(device="cuda", distributed=False); tr.train()
reals = synthetic_batch(B, "cuda")
for _ in range(3):
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
rows = []


def stat(kind, role, t, m, n, k, by_col=False):
    t = t.detach()
    mx = t.abs().max()
    rms = t.pow(2).mean().sqrt()
    tot = t.pow(2).sum().clamp_min(1e-300)
    unit = t.abs().amax(0 if by_col else 1, keepdim=True)            # the scaling unit's maximum: per column (transposed operand) / per row
    small = t.abs() < unit * 2.0 ** -16
    old = t.abs() < mx * 2.0 ** -16
    rows.append((kind, role, "col" if by_col else "row", m, n, k, float(mx), float(torch.log2(mx / rms.clamp_min(1e-300))),
                 float(torch.log2(mx / unit.clamp_min(1e-300).min().clamp_min(1e-300))), float(small.float().mean()),
                 float((t * small).pow(2).sum() / tot), float(old.float().mean()), float((t * old).pow(2).sum() / tot)))


orig_planes, orig_nt, orig_tn = fused.gemm_nt_planes, fused.gemm_nt, fused.gemm_tn


def planes(a, P, n, k, *args, **kw):
    if P.shape[0] == 2:
        stat("nt (planes)", "A", a, a.shape[0], n, k)
    return orig_planes(a, P, n, k, *args, **kw)


def nt(a, w, *args, **kw):
    wt = kw.get("w_transposed", False)
    n = w.shape[1] if wt else w.shape[0]
    if fused.two_part(a.shape[0], (n + 3) // 4 * 4, (a.shape[1] + 3) // 4 * 4, 0):
        stat("nn" if wt else "nt", "A", a, a.shape[0], n, a.shape[1])
        stat("nn" if wt else "nt", "W", w, a.shape[0], n, a.shape[1], by_col=wt)
    return orig_nt(a, w, *args, **kw)


def tn(dy, x, *args, **kw):
    if fused.two_part(dy.shape[1], x.shape[1], dy.shape[0], 0):
        stat("tn", "dY", dy, dy.shape[0], dy.shape[1], x.shape[1], by_col=True)
        stat("tn", "X", x, dy.shape[0], dy.shape[1], x.shape[1], by_col=True)
    return orig_tn(dy, x, *args, **kw)


fused.gemm_nt_planes, fused.gemm_nt, fused.gemm_tn = planes, nt, tn
tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
print("kind          role unit      m      n      k        max  log2(max/rms)  log2(max/smallest unit max)   < 2^-16 of unit max: elements  energy | "
      "of the operand's max (round 5): elements  energy")
for r in rows:
    print("%-12s %-4s %-4s %7d %6d %6d  %9.3e  %8.1f  %12.1f  %24.2e  %10.2e | %20.2e  %10.2e" % r)
