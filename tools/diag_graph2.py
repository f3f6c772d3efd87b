import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
mode = sys.argv[1]
B = 35
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
if "comm" in mode:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
tr = PDGNTrainer(device="cuda", distributed=True); tr.train()
if "comm" not in mode:
    tr._comm = lambda k: None
reals = synthetic_batch(B, "cuda")
zs = [(noise(B, "cuda"), noise(B, "cuda")) for _ in range(8)]
tr.capture(reals, *zs[0], warmup=2); torch.cuda.synchronize()
print("captured", len(tr._graphs), "graphs", flush=True)
sync = torch.cuda.synchronize if "devsync" in mode else (lambda: torch.cuda.current_stream().synchronize())
for it in range(8):
    st = tr._static
    sync(); st["z1"].copy_(zs[it][0]); st["z2"].copy_(zs[it][1]); sync()
    for g, last in tr._graphs:
        g.replay()
        if "nosync" not in mode: sync()
        if "comm" in mode and last < 5:
            tr._comm(last); sync()
torch.cuda.synchronize()
print("mode", mode, "ok", {k: round(v.item(), 4) for k, v in tr._static["out"].items()}, flush=True)
