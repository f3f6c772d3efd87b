"""How far ahead may the host run with the launch list?  (back-to-back vs bounded lead)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = 35
dev = torch.device("cuda", 0)
reals = synthetic_batch(B, dev)
gen = torch.Generator().manual_seed(5)
zs = [(noise(B, dev, gen), noise(B, dev, gen)) for _ in range(12)]
torch.manual_seed(9999)
tr = PDGNTrainer(device=dev, distributed=False)
tr.train()
for i in range(3):
    tr.step(reals, *zs[i])
tr.capture_list(reals, *zs[0])
for i in range(3):
    tr.step_list(None, *zs[i])
torch.cuda.synchronize()
N = 30


def run(lead):
    evs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        if lead is not None and len(evs) >= lead:
            evs[-lead].synchronize()
        tr.step_list(None, *zs[i % 10])
        e = torch.cuda.Event()
        e.record()
        evs.append(e)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / N * 1e3, (t1 - t0) / N * 1e3


for rep in range(2):
    for lead in (None, 1, 2, 3):
        ms, iss = run(lead)
        print("lead %-5s %.2f ms/step (host loop %.2f ms/step)" % (lead, ms, iss))
t0 = time.perf_counter()
for i in range(N):
    tr.step(reals, *zs[i % 10])
torch.cuda.synchronize()
print("eager %.2f ms/step" % ((time.perf_counter() - t0) / N * 1e3))
