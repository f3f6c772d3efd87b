#!/usr/bin/env python3
"""The generator's Adam step alone: torch._fused_adam_ (through trainer.LeanAdamStep, PDGN_OWN_ADAM=0) against csrc/adam.hip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.generator import PointGenerator
from pdgn_amd.trainer import LeanAdamStep
G = PointGenerator().cuda()
ps = [p for p in G.parameters()]
print("tensors %d, elements %.2f M" % (len(ps), sum(p.numel() for p in ps) / 1e6))
for own in (False, True):
    opt = torch.optim.Adam(ps, lr=1e-4, betas=(0.5, 0.999), fused=True, capturable=True)
    lean = LeanAdamStep(opt)
    lean._OWN = own
    for p in ps:
        p.grad = torch.randn_like(p) * 1e-3
    for _ in range(3):
        lean.step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        lean.step()
    e.record(); torch.cuda.synchronize()
    print("own kernel" if own else "torch fused", "%.1f us per step" % (s.elapsed_time(e) / 20 * 1e3))
