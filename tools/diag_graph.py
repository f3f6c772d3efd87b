import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
z1, z2 = noise(B, "cuda"), noise(B, "cuda")
print("eager step", flush=True)
tr.step(reals, z1, z2); torch.cuda.synchronize()
print("capture", flush=True)
tr.capture(reals, *( (None, None) if "devnoise" in sys.argv else (z1, z2)), warmup=1); torch.cuda.synchronize()
print("captured; mem GB", torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9, flush=True)
for it in range(3):
    for g, k in tr._graphs:
        g.replay(); torch.cuda.synchronize()
        print("iter", it, "segment", k, "ok", flush=True)
print({k: v.item() for k, v in tr._static["out"].items()})
mode = sys.argv[2] if len(sys.argv) > 2 else "pure"
print("mode", mode, flush=True)
zs = [(noise(B, "cuda"), noise(B, "cuda")) for _ in range(10)]
torch.cuda.synchronize()
for it in range(10):
    if mode == "devnoise":
        tr.step_graphed()
    elif mode == "pure":
        for g, k in tr._graphs: g.replay()
    elif mode == "copy":
        tr.step_graphed(reals, *zs[it])
    elif mode == "copysync":
        tr.step_graphed(reals, *zs[it]); torch.cuda.synchronize()
    elif mode == "h2d":
        tr.step_graphed(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
print("mode", mode, "ok", {k: round(v.item(), 4) for k, v in tr._static["out"].items()}, flush=True)
