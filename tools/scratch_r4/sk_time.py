import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pdgn_amd import fused
dev = torch.device("cuda", 0)
B = 35
for _ in range(20):
    for (Fc, Mw) in ((32, 3232), (64, 6432), (128, 12832)):
        c = torch.randn(B, Fc, device=dev); W = torch.randn(Mw, Fc, device=dev); g = torch.randn(B, Mw, device=dev)
        fused.skinny_nt(c, W); fused.skinny_nn(g, W); fused.skinny_tn(g, c)
    g = torch.randn(B, 512, device=dev); W0 = torch.randn(256, 768, device=dev); b0 = torch.randn(256, device=dev); drb = torch.randn(B, 256, device=dev)
    fused.skinny_nt(g, W0[:, :512], b0); fused.skinny_tn(drb, g); fused.skinny_nn(drb, W0[:, :512])
    for (N, K) in ((512, 1024), (256, 512), (4096, 128)):
        dp = torch.randn(B, N, device=dev); W = torch.randn(N, K, device=dev)
        fused.skinny_nn(dp, W)
torch.cuda.synchronize()
