"""Issue order of the launch list (PDGN_REPLAY_BURST) x stream layout, pace 1.0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
from pdgn_amd import streams, replay
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
graph = tr._list.graph
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
    for burst in ("0", "1000000", "16", "6", "3", "1"):
        os.environ["PDGN_REPLAY_BURST"] = burst
        tr._list = replay.LaunchList(graph)
        print("burst %-8s %.2f ms/step" % (burst, run()), flush=True)
os.environ["PDGN_REPLAY_BURST"] = os.environ.get("BEST_BURST", "0")
tr._list = replay.LaunchList(graph)
for layout in ("BCBCAA", "BBBCAA", "BCBCAA", "BBBCAA", "ABBCAA", "BBCCAA"):
    os.environ["PDGN_STREAM_LAYOUT"] = layout
    streams.reset()
    tr._list._bound = None
    print("layout %s  %.2f ms/step" % (layout, run()), flush=True)
