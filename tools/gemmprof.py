#!/usr/bin/env python3
"""Shape-resolved GEMM time of one G+D step (aten::mm / addmm / bmm with input shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
for _ in range(3):
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
    torch.cuda.synchronize()
rows = []
tot = 0.0
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::mm", "aten::addmm", "aten::bmm"):
        sh = [s for s in e.input_shapes if s]
        t = e.self_device_time_total / 1e3
        tot += t
        if e.key == "aten::addmm":
            m, k = sh[1]; k2, n = sh[2]
        elif e.key == "aten::bmm":
            bb, m, k = sh[0]; _, k2, n = sh[1]; m *= bb
        else:
            m, k = sh[0]; k2, n = sh[1]
        fl = 2.0 * m * k * n * e.count
        rows.append((t, e.key, e.count, sh, fl / (t * 1e-3) / 1e12 if t > 0 else 0))
rows.sort(key=lambda r: -r[0])
print("total GEMM ms", tot)
for t, k, c, sh, tf in rows[:45]:
    print("%7.3f ms  x%-3d %-11s %6.1f TF  %s" % (t, c, k, tf, sh))
