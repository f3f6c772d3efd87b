#!/usr/bin/env python3
"""How long does the default stream wait for the feature-kNN graphs inside the overlapped step?  (events around the
wait_event calls of PointDeconv.forward_cl; no tracer)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B, STEPS = 35, 10
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
zs = [(noise(B, "cuda"), noise(B, "cuda")) for _ in range(STEPS + 5)]
for i in range(5):
    tr.step(reals, *zs[i])
torch.cuda.synchronize()
log = []
orig = torch.cuda.Stream.wait_event
main = torch.cuda.current_stream()
def patched(self, ev):
    if self.cuda_stream != main.cuda_stream:
        return orig(self, ev)
    e0 = torch.cuda.Event(enable_timing=True); e0.record(self)
    orig(self, ev)
    e1 = torch.cuda.Event(enable_timing=True); e1.record(self)
    log.append((e0, e1))
torch.cuda.Stream.wait_event = patched
for i in range(STEPS):
    tr.step(reals, *zs[5 + i])
torch.cuda.synchronize()
torch.cuda.Stream.wait_event = orig
per = len(log) // STEPS
tot = [0.0] * per
for i, (a, b) in enumerate(log):
    tot[i % per] += a.elapsed_time(b)
print("default-stream wait_event calls per step: %d" % per)
print("mean wait per call (ms):", " ".join("%.3f" % (t / STEPS) for t in tot))
print("total %.3f ms/step" % (sum(tot) / STEPS))
