import os, sys
sys.path.insert(0, "/root/repo")
import torch
from torch.profiler import profile, ProfilerActivity
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B, dev = 35, torch.device("cuda", 0)
torch.manual_seed(9999)
tr = PDGNTrainer(device=dev, distributed=False)
tr.train()
reals = synthetic_batch(B, dev)
g = torch.Generator().manual_seed(1234)
zs = [(noise(B, dev, g), noise(B, dev, g)) for _ in range(4)]
for i in range(3):
    tr.step(reals, *zs[i])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.step(reals, *zs[3])
    torch.cuda.synchronize()
from collections import Counter
c = Counter()
ex = {}
for e in prof.events():
    if e.name.startswith("aten::") and e.name.split("::")[1] in ("sum", "add", "add_", "fill_", "zero_", "mm", "addmm", "matmul", "mul", "mul_", "copy_", "cat", "stack", "div", "sub", "neg", "where", "clone", "contiguous", "zeros", "_foreach_add_", "mean", "index_select", "gather"):
        key = (e.name, str(e.input_shapes)[:90])
        c[key] += 1
        if key not in ex and e.stack:
            ex[key] = [f for f in e.stack if "pdgn_amd" in f][:2]
for (n, sh), k in c.most_common(60):
    print("%3d  %-16s %s   %s" % (k, n, sh, "; ".join(x.split("/")[-1][:60] for x in ex.get((n, sh), []))))
