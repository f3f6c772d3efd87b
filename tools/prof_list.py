#!/usr/bin/env python3
"""The launch-list step alone (no roofline / eval / baseline work), for rocprofv3 traces: python3 tools/prof_list.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, dev = 35, torch.device("cuda", 0)
torch.manual_seed(9999)
tr = PDGNTrainer(device=dev, distributed=False)
tr.train()
reals = synthetic_batch(B, dev)
g = torch.Generator().manual_seed(1234)
zs = [(noise(B, dev, g), noise(B, dev, g)) for _ in range(4)]
for i in range(3):
    tr.step(reals, *zs[i])
if os.environ.get("EAGER") == "1":
    step = lambda i: tr.step(reals, *zs[i % 4])
else:
    tr.capture_list(reals, *zs[0])
    step = lambda i: tr.step_list(None, *zs[i % 4])
for i in range(steps + 3):
    step(i)
torch.cuda.synchronize()
