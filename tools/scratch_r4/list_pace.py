"""Launch-list pacing (fraction of the previous iteration's main-stream launches to wait for) and stream layouts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
from pdgn_amd import streams
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
N = 30


def run():
    for i in range(3):
        tr.step_list(None, *zs[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        tr.step_list(None, *zs[i % 10])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


for rep in range(2):
    for pace in (1.0, 0.9, 0.75, 0.6, 0.45, 0.3):
        tr._list_pace = pace
        print("pace %.2f  %.2f ms/step" % (pace, run()), flush=True)
best = float(os.environ.get("BEST_PACE", "0.6"))
tr._list_pace = best
for layout in ("BCBCAA", "BBBCAA", "BBCCAA", "BCCBAA", "ABBCAA", "BBBCAC"):
    os.environ["PDGN_STREAM_LAYOUT"] = layout
    streams.reset()
    tr._list._bound = None
    print("layout %s  %.2f ms/step (pace %.2f)" % (layout, run(), best), flush=True)
