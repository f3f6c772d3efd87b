#!/usr/bin/env python3
"""Host issue time vs device time of the step: is the eager schedule CPU- or GPU-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = 35
force = os.environ.get("PDGN_FORCE_DIST") == "1"           # the RCCL path on one GPU (world_size 1)
if force:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    torch.cuda.set_device(0)
    pg = os.environ.get("PDGN_X_PG", "nccl")
    if pg == "nccl":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    elif pg == "nccl-lazy":
        dist.init_process_group("nccl", rank=0, world_size=1)
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)
tr = PDGNTrainer(device="cuda", distributed=(os.environ.get("PDGN_X_DIST", "1") == "1") if force else None); tr.train()
reals = synthetic_batch(B, "cuda")
zs = [(noise(B, "cuda"), noise(B, "cuda")) for _ in range(30)]
for i in range(5):
    tr.step(reals, *zs[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    tr.step(reals, *zs[5 + i])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue %.2f ms/step, total %.2f ms/step, drain after last issue %.2f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, (t2 - t1) * 1e3))
