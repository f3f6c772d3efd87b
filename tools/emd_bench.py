#!/usr/bin/env python3
"""Fused EMD cost (pdgn_emd_cost) at the C5 evaluation shape: 512 pairs of 2048 x 2048 points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.structural_losses import emd_cost
B, N = 512, 2048
g = torch.Generator().manual_seed(9999)
a = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda(); b = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
for _ in range(2): c = emd_cost(a, b)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): c = emd_cost(a, b)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
# 9 levels x 3 phases of n*m exp evaluations per pair: 113.2 M per pair
print("emd_cost %d pairs of %dx%d: %.2f ms  (%.1f G exp/s; checksum %.6f)" % (B, N, N, ms, B * 27 * N * N / ms / 1e6, c.double().sum().item()))
