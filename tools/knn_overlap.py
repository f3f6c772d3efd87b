"""How much do the feature-kNN neighbourhoods of CONSECUTIVE points overlap?  (decides whether a gather-sum that stages the union of a
group's neighbour rows in LDS could cut the texture-path requests of wgs_fwd_*: DESIGN.md section 10b)
For each block of the generator (random-init weights, B = 8): unique neighbour rows of G consecutive points / (G * k)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import deconv
from pdgn_amd.generator import PointGenerator
from pdgn_amd.trainer import noise

dev = torch.device("cuda:0")
torch.manual_seed(9999)
G = PointGenerator().to(dev).train()
graphs = []
orig = deconv.feature_knn


def spy(x, k, *a, **kw):
    idx = orig(x, k, *a, **kw)
    graphs.append(idx.detach().clone())
    return idx


deconv.feature_knn = spy
if hasattr(deconv, "start_feature_knn"):
    o2 = deconv.start_feature_knn

    def spy2(*a, **kw):
        out = o2(*a, **kw)
        graphs.append(out[0].detach().clone() if isinstance(out, tuple) else out.detach().clone())
        return out
    deconv.start_feature_knn = spy2
with torch.no_grad():
    for _ in range(3):                       # a few passes so that BatchNorm statistics / the lists settle
        graphs.clear()
        G(noise(8, dev))
torch.cuda.synchronize()
for idx in graphs:
    B, N, k = idx.shape
    line = "N %5d k %2d:" % (N, k)
    for grp in (8, 16, 32, 64):
        g = idx.view(B, N // grp, grp * k).long()
        uniq = torch.tensor([[len(torch.unique(g[b, j])) for j in range(g.shape[1])] for b in range(B)], dtype=torch.float32)
        line += "  G=%2d unique/(G k) %.2f" % (grp, uniq.mean().item() / (grp * k))
    print(line)
