#!/usr/bin/env python3
"""Per-phase GPU time of one G+D step (CUDA events around the phases of PDGNTrainer.step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
ones = torch.ones(B, 1, device="cuda"); zeros = torch.zeros(B, 1, device="cuda")
for _ in range(3):
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
acc = {}
for it in range(5):
    marks.clear()
    z1, z2 = noise(B, "cuda"), noise(B, "cuda")
    torch.cuda.synchronize(); mark("start")
    with torch.no_grad():
        fakes = tr.G(z1)
    mark("G fwd #1 (no grad)")
    for i, D in enumerate(tr.D):
        tr.gradD[i].begin()
        lossD = (F.mse_loss(D(reals[i]), ones) + F.mse_loss(D(fakes[i]), zeros)) / 2.0
        lossD.backward(); tr.optD[i].step()
    mark("4x D step")
    tr.gradG.begin(); tr._freeze_D(True)
    gen = tr.G(z2)
    mark("G fwd #2 (grad)")
    sim = tr.similar_loss(gen)
    mark("local-pair loss fwd")
    g_loss = [F.mse_loss(tr.D[i](gen[i]), ones) for i in range(4)]
    lossG = 1.2 * g_loss[0] + 1.2 * g_loss[1] + 1.2 * g_loss[2] + g_loss[3] + 0.1 * sim
    mark("D(gen) fwd")
    lossG.backward()
    mark("backward (D + local-pair + G)")
    tr._freeze_D(False); tr.optG.step()
    mark("Adam G")
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1) / 5
tot = sum(acc.values())
for k, v in acc.items():
    print("%-34s %8.2f ms  %5.1f%%" % (k, v, 100 * v / tot))
print("%-34s %8.2f ms" % ("total", tot))
# backward split: per-stage deconv backward via hooks is intrusive; instead time G fwd per block
G = tr.G
with torch.no_grad():
    x = G.fc1(z1).view(B, 32, G.base_points)
    blocks = [("bilateral1", lambda: G.bilateral1(x))]
    torch.cuda.synchronize()
    def t(fn, n=5):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): r = fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n, r
    ms, (x1, g1) = t(lambda: G.bilateral1(x)); print("fwd bilateral1 %.2f ms" % ms)
    p1 = G.mlp1(g1)
    ms, (x2, g2) = t(lambda: G.bilateral2(x1, p1)); print("fwd bilateral2 %.2f ms" % ms)
    p2 = G.mlp2(g2)
    ms, (x3, g3) = t(lambda: G.bilateral3(x2, p2)); print("fwd bilateral3 %.2f ms" % ms)
    p3 = G.mlp3(g3)
    ms, x4 = t(lambda: G.bilateral4(x3, p3)); print("fwd bilateral4 %.2f ms" % ms)
    ms, _ = t(lambda: G.mlp4(x4)); print("fwd mlp4 %.2f ms" % ms)
    ms, _ = t(lambda: (G.mlp1(g1), G.mlp2(g2), G.mlp3(g3))); print("fwd mlp1-3 %.2f ms" % ms)
    for i, D in enumerate(tr.D):
        ms, _ = t(lambda: D(reals[i])); print("fwd D%d %.2f ms" % (i + 1, ms))
