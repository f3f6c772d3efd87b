"""Where does the one-rank RCCL launch list die?  (measurement)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sleep = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
tr = PDGNTrainer(device=dev, distributed=True); tr.train()
B = 35
reals, z1, z2 = synthetic_batch(B, dev), noise(B, dev), noise(B, dev)
for _ in range(2): tr.step(reals, z1, z2)
torch.cuda.synchronize(); print("eager ok", flush=True)
if sleep: time.sleep(sleep)
tr.capture_list(reals, z1, z2)
torch.cuda.synchronize(); print("captured", tr._list.info, len(tr._list_points), flush=True)
time.sleep(1.0); print("slept after capture", flush=True)
for i in range(20):
    tr.step_list(None, z1, z2)
    if i % 5 == 0:
        torch.cuda.synchronize(); print("replay", i, flush=True)
torch.cuda.synchronize(); time.sleep(1.0); print("done", flush=True)
tr._list = None
dist.destroy_process_group()
