import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from collections import Counter
from pdgn_amd import fused
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
tr = PDGNTrainer(device="cuda", distributed=False); tr.train()
reals = synthetic_batch(35, "cuda")
for _ in range(3):
    tr.step(reals, noise(35, "cuda"), noise(35, "cuda"))
torch.cuda.synchronize()
orig = fused._zeros
log = Counter()
def z(shape, device):
    a = fused._ARENA
    n = 1
    for d in shape: n *= d
    step = (n + 63) // 64 * 64
    buf = a["buf"]
    why = "ok"
    if buf is None: why = "nobuf"
    elif buf.device != device: why = "dev"
    elif a["off"] + step > buf.numel(): why = "full(%d+%d>%d)" % (a["off"], step, buf.numel())
    log[(type(a["owner"]).__name__ + str(id(a["owner"]) % 1000), why if why[:4] != "full" else "full")] += 1
    if why[:4] == "full" and log[("x", "printed")] < 5:
        log[("x", "printed")] += 1
        print(why, shape)
    return orig(shape, device)
fused._zeros = z
import pdgn_amd.deconv as dc
dc._zeros = z
tr.step(reals, noise(35, "cuda"), noise(35, "cuda"))
torch.cuda.synchronize()
for k, v in log.items(): print(k, v)
print([ (getattr(g, "zero_arena_floats", None)) for g in tr.gradD + [tr.gradG]])
