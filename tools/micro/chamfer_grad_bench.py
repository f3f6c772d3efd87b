#!/usr/bin/env python3
"""Isolated timings of the Chamfer adjoint (losses.chamfer_sum backward) at the local-pair sizes; run under rocprofv3 --kernel-trace --stats
with PDGN_CHAMFER_LDS=0 / 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdgn_amd import losses
B = 35
for m, d in ((256, 3), (256, 9), (512, 9), (1024, 3), (1024, 9)):
    x = torch.randn(B, m, d, device="cuda").requires_grad_(True)
    y = torch.randn(B, m, d, device="cuda").requires_grad_(True)
    for _ in range(20):
        losses.chamfer_sum(x, y, 0.25).backward()
torch.cuda.synchronize()
