#!/usr/bin/env python3
"""How long does one no-grad generator pass take on the issuing stream and on each side stream of the plan (alone on the device)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import deconv, streams as _streams
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B, dev = 35, torch.device("cuda", 0)
torch.manual_seed(9999)
tr = PDGNTrainer(device=dev); tr.train()
reals = synthetic_batch(B, dev)
z = noise(B, dev)
for _ in range(2):
    tr.step(reals, noise(B, dev), noise(B, dev))
pl = _streams.plan(dev)
main = torch.cuda.current_stream(dev)
tr.G.preassemble()
def run(s, inline):
    old = deconv._KNN_OVERLAP
    deconv._KNN_OVERLAP = not inline
    try:
        for rep in range(3):
            s.wait_stream(main)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            with torch.cuda.stream(s), torch.no_grad():
                e0.record(s)
                tr.G(z)
                e1.record(s)
            torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    finally:
        deconv._KNN_OVERLAP = old
print("main, kNN overlapped", run(main, False), " inline", run(main, True))
for name, s in (("D1", pl.d[0]), ("D4", pl.d[3]), ("lp", pl.lp), ("knn", pl.knn)):
    print(name, "inline kNN", run(s, True))
