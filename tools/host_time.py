#!/usr/bin/env python3
"""Host issue time vs device time of the step: is the eager schedule CPU- or GPU-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
zs = [(noise(B, "cuda"), noise(B, "cuda")) for _ in range(30)]
for i in range(5):
    tr.step(reals, *zs[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    tr.step(reals, *zs[5 + i])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue %.2f ms/step, total %.2f ms/step, drain after last issue %.2f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, (t2 - t1) * 1e3))
