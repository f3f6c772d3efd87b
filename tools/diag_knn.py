import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import numpy as np, torch
from hashweights import fill_module
from pdgn_amd.generator import PointGenerator
from pdgn_amd import deconv
g = dict(np.load("tests/golden/generator_b6.npz"))
G = fill_module(PointGenerator(), salt=1).cuda().train()
got = []
orig = deconv.feature_knn
def spy(x, k):
    i = orig(x, k); got.append((x.detach().cpu(), i.cpu())); return i
deconv.feature_knn = spy
# feed golden graphs to keep inputs identical, but ALSO compute own graph on the same inputs
blocks = [G.bilateral1.upsample_cov[0], G.bilateral2.upsample_cov, G.bilateral3.upsample_cov, G.bilateral4.upsample_cov]
def hook(m, inp):
    spy(inp[0].contiguous(), m.k)
for b in blocks:
    b.register_forward_pre_hook(hook)
with torch.no_grad():
    G(torch.from_numpy(g["z"]).cuda(), idx=[torch.from_numpy(g["idx%d" % i].astype(np.int32)).cuda() for i in (1,2,3,4)])
for s, (x, idx) in enumerate(got):
    gold = torch.from_numpy(g["idx%d" % (s+1)].astype(np.int64))
    same = (idx.long() == gold).all(2)
    xt = x.double().transpose(1,2)
    d = -2*torch.bmm(xt, x.double()) + (xt**2).sum(2,keepdim=True) + (xt**2).sum(2).unsqueeze(1)
    ds = d.sort(2)[0][:, :, :12]
    gap = (ds[:, :, 1:]-ds[:, :, :-1]).amin(2) / d.abs().amax(2)
    print("stage", s+1, "rows", same.numel(), "mismatch rows", int((~same).sum()), "min rel gap on mismatches", gap[~same].max().item() if (~same).any() else None,
          "frac rows with gap<1e-6", (gap < 1e-6).float().mean().item(), "golden margin", g["knn_margins"][s])
    # set-level agreement
    a = idx.long().sort(2)[0]; b = gold.sort(2)[0]
    print("   set mismatch rows", int((a != b).any(2).sum()))
