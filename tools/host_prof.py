#!/usr/bin/env python3
"""cProfile of the host side of the step (which Python / dispatcher work the ~2500 launches cost)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
zs = [(noise(B, "cuda"), noise(B, "cuda")) for _ in range(20)]
for i in range(5):
    tr.step(reals, *zs[i])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    tr.step(reals, *zs[5 + i])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
st.sort_stats("cumulative").print_stats(45)
