#!/usr/bin/env python3
"""Losses of 8 eager iterations from a fixed seed: python3 tools/traj_check.py (with / without PDGN_CLOSED_TAIL=0 PDGN_STATS_MAX=0 ...).
Iteration 1 agrees to 1e-5 between the arithmetic variants; from iteration 2 on the runs differ by ~1e-2 -- as much as two runs of the SAME
variant differ from each other (float atomics in the weight gradients / kNN ties flip neighbours): the dynamics, not the variant."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B, dev = 35, torch.device("cuda", 0)
torch.manual_seed(9999)
tr = PDGNTrainer(device=dev, distributed=False)
tr.train()
reals = synthetic_batch(B, dev)
g = torch.Generator().manual_seed(1234)
zs = [(noise(B, dev, g), noise(B, dev, g)) for _ in range(8)]
out = []
for i in range(8):
    o = tr.step(reals, *zs[i])
    out.append([float(o[k]) for k in ("d_loss1", "d_loss2", "d_loss3", "d_loss4", "g_loss", "similar_loss")])
for r in out:
    print(" ".join("%.5f" % v for v in r))
