#!/usr/bin/env python3
"""Torch ops (not the package's kernels) launched by the autograd thread during one step's backward, by op and shape."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import defaultdict
from torch.profiler import profile, ProfilerActivity
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
for _ in range(3):
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
main_tid = threading.get_native_id()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
tids = defaultdict(int)
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    dt = sum(k.duration for k in e.kernels)
    if dt <= 0 or any(c.name.startswith("aten::") and sum(k.duration for k in c.kernels) > 0 for c in e.cpu_children):
        continue                                              # only the leaf op that owns the kernels
    tids[e.thread] += 1
    key = (e.thread, e.name, str(e.input_shapes)[:90])
    agg[key][0] += len(e.kernels)
    agg[key][1] += dt
bw = [t for t in tids if t != main_tid]
print("threads:", dict(tids), "main", main_tid)
rows = sorted(((v[0], v[1], k) for k, v in agg.items() if k[0] in bw), key=lambda r: -r[0])
print("autograd-thread torch ops: %d launches, %.2f ms" % (sum(r[0] for r in rows), sum(r[1] for r in rows) / 1e3))
for n, t, k in rows[:70]:
    print("%4d  %8.1f us  %-26s %s" % (n, t, k[1], k[2]))
