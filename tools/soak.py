#!/usr/bin/env python3
"""Soak: N iterations of the overlapped step; losses stay finite, allocator pools stop growing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
g = torch.Generator().manual_seed(0)
t0 = time.time()
for it in range(N):
    out = tr.step(reals, noise(B, "cuda", g), noise(B, "cuda", g))
    if it % 25 == 0 or it == N - 1:
        vals = {k: round(float(v), 4) for k, v in out.items()}
        assert all(v == v and abs(v) < 1e6 for v in vals.values()), vals
        print("it %4d  %.1f s  reserved %.2f GB  allocated %.2f GB  %s" % (
            it, time.time() - t0, torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30, vals), flush=True)
print("soak ok")
