import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.fused import bn_act
for rows, C in ((179200, 64), (35840, 512)):
    for shift in (0, 3, 10, 30, 100, 300, 1000):
        g = torch.Generator(device="cuda").manual_seed(rows + shift)
        x = torch.randn(rows, C, device="cuda", generator=g) + shift
        bn = torch.nn.BatchNorm1d(C).cuda().train()
        y = bn_act(x, bn, True, act="none")
        ref = torch.nn.functional.batch_norm(x.double(), None, None, bn.weight.double(), bn.bias.double(), True, 0.1, bn.eps)
        yt = torch.nn.functional.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.1, bn.eps)
        err = (y.double() - ref).abs().max().item()
        errt = (yt.double() - ref).abs().max().item()
        print("rows %6d C %3d  mean/std %5d:  max |y - fp64| = %.2e   (torch fp32: %.2e)" % (rows, C, shift, err, errt), flush=True)
