#!/usr/bin/env python3
"""ms per iteration of the tree this file lives in (launch list when the tree has it, else eager): python3 tools/step_time.py [steps]
For A/B runs of two trees on ONE box (tools/ab_trees.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, dev = 35, torch.device("cuda", 0)
torch.manual_seed(9999)
tr = PDGNTrainer(device=dev, distributed=False)
tr.train()
reals = synthetic_batch(B, dev)
g = torch.Generator().manual_seed(1234)
zs = [(noise(B, dev, g), noise(B, dev, g)) for _ in range(6)]
for i in range(3):
    tr.step(reals, *zs[i])
mode = "eager"
step = lambda i: tr.step(reals, *zs[i % 6])
if hasattr(tr, "capture_list") and os.environ.get("EAGER") != "1":
    tr.capture_list(reals, *zs[0])
    step = lambda i: tr.step_list(None, *zs[i % 6])
    mode = "list"
for i in range(5):
    step(i)
torch.cuda.synchronize()
res = []
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / steps * 1e3)
print("%s %s  %.2f %.2f ms/step" % (os.path.basename(ROOT) or ROOT, mode, res[0], res[1]), flush=True)
