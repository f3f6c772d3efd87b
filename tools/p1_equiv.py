#!/usr/bin/env python3
"""One iteration with generator pass #1 concurrent on D4's stream (PDGN_PASS1_SIDE=1, the default) against the same iteration with both
passes on the issuing stream, from identical state: losses, parameters, BatchNorm buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B, dev = int(os.environ.get("B", "8")), torch.device("cuda", 0)
res = {}
for side in (True, False):
    torch.manual_seed(9999)
    tr = PDGNTrainer(device=dev, distributed=False); tr.train()
    tr._pass1_side = side
    g = torch.Generator().manual_seed(5)
    reals = synthetic_batch(B, dev, seed=3)
    outs = []
    for it in range(3):
        z1, z2 = noise(B, dev, g), noise(B, dev, g)
        outs.append({k: float(v) for k, v in tr.step(reals, z1, z2).items()})
    torch.cuda.synchronize()
    res[side] = (outs, [p.detach().clone() for net in [tr.G] + tr.D for p in net.parameters()],
                 [b.detach().clone().float() for net in [tr.G] + tr.D for b in net.buffers()])
for it in range(3):
    print("iteration", it)
    for k in sorted(res[True][0][it]):
        print("  %-14s side %.6f  main %.6f" % (k, res[True][0][it][k], res[False][0][it][k]))
dp = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)) for a, b in zip(res[True][1], res[False][1]))
db = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)) for a, b in zip(res[True][2], res[False][2]))
print("largest relative difference: parameters %.3e  buffers %.3e" % (dp, db))
