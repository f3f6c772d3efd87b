#!/usr/bin/env python3
"""What the data-parallel path adds per iteration at world size 1 (measurement): the generator's gradient pack (multi-tensor copy into
the flat buffer), its all-reduce on a one-rank RCCL communicator, and the Adam step on the flat buffer's views, each alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
tr = PDGNTrainer(device=dev, distributed=True); tr.train()
B = 35
reals, z1, z2 = synthetic_batch(B, dev), noise(B, dev), noise(B, dev)
for _ in range(2): tr.step(reals, z1, z2)
torch.cuda.synchronize()
fg = tr.gradG


def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


fresh = [torch.randn_like(p) for p in fg.params]
def pack():
    for p, g in zip(fg.params, fresh): p.grad = g
    fg._early_done = False
    fg.pack()
print("pack (multi-tensor copy of %d tensors, %.1f MB): %.1f us" % (len(fg.params), fg.buf.numel() * 4 / 1e6, t(pack)))
print("all-reduce of the flat buffer, one rank: %.1f us" % t(lambda: dist.all_reduce(fg.buf)))
print("all-reduce early bucket + rest: %.1f us" % t(lambda: (dist.all_reduce(fg.buf[:fg.early_numel]), dist.all_reduce(fg.buf[fg.early_numel:]))))
pack()
print("Adam step on the views: %.1f us" % t(tr._stepG.step))
dist.destroy_process_group()
