import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests")]
import numpy as np, torch
from hashweights import fill_module, hash_tensor
from torch_standins import feature_knn_torch
from pdgn_amd import deconv
from pdgn_amd.trainer import PDGNTrainer
g = dict(np.load(os.path.join(R, "tests/golden/step_b4.npz")))
B = 4
def run(tag, knn=None, dtype=torch.float32):
    orig = deconv.feature_knn
    if knn is not None:
        deconv.feature_knn = knn
    tr = PDGNTrainer(device="cuda", distributed=False)
    fill_module(tr.G, salt=1)
    for i, d in enumerate(tr.D):
        fill_module(d, salt=10 + i)
    tr.train()
    reals = [hash_tensor("real%d" % i, (B, 3, n), 0.8).cuda() for i, n in enumerate((256, 512, 1024, 2048))]
    out = tr.step(reals, hash_tensor("step_z1", (B, 128), 0.2).cuda(), hash_tensor("step_z2", (B, 128), 0.2).cuda())
    deconv.feature_knn = orig
    print(tag, {k: "%.6f (gold %.6f, rel %.1e)" % (v.item(), g[k], abs(v.item() - g[k]) / abs(g[k])) for k, v in out.items()})
run("hip-knn")
def knn64(x, k):
    return feature_knn_torch(x.double(), k)
run("torch-fp64-knn", knn64)
run("torch-fp32-knn", feature_knn_torch)
