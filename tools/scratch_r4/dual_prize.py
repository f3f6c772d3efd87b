"""What would batching the two generator passes through blocks 1-3 buy?  Forward of blocks 1-3 (no autograd) at B = 35 twice vs
B = 70 once, and block 4 at B = 35, each as a launch list (no host in the way)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pdgn_amd import replay, streams
from pdgn_amd.generator import PointGenerator
from pdgn_amd.trainer import noise
dev = torch.device("cuda", 0)
torch.manual_seed(1)
G = PointGenerator().to(dev).train()
keep = []


def build(B, levels):
    z = noise(B, dev)

    def run():
        with torch.no_grad():
            s = G.begin(z)
            for lvl in range(levels):
                G.level(s, lvl)
            return s["xt"]
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        main = torch.cuda.current_stream()
        pl = streams.plan(dev)
        replay.mark(replay.MAIN, main)
        pl.knn.wait_stream(main)
        replay.mark(replay.KNN, pl.knn)
        main.wait_stream(pl.knn)
        out = run()
        main.wait_stream(pl.knn)
    ll = replay.LaunchList(g)
    keep.append((g, ll, out))
    return ll


def timeit(ll, reps=1, iters=30):
    pl = streams.plan(dev)
    ll.bind({replay.MAIN: torch.cuda.current_stream(), replay.KNN: pl.knn}, [torch.cuda.Stream() for _ in range(2)])
    for _ in range(3):
        for _ in range(reps):
            ll.launch()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for _ in range(reps):
            ll.launch()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for levels in (1, 2, 3, 4):
    a, b = build(35, levels), build(70, levels)
    print("blocks 1..%d: B=35 once %.3f ms, B=35 twice %.3f ms, B=70 once %.3f ms  (%s)" % (
        levels, timeit(a), timeit(a, reps=2), timeit(b), a.info), flush=True)
sys.stdout.flush()
os._exit(0)
