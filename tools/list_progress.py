#!/usr/bin/env python3
"""Device-time progress of the issuing stream's chain through ONE launch-list iteration, untraced (timing events after every
N-th launch of the chain; csrc/replay.hip::pdgn_replay_probe_chain), next to the same chain's kernels in a rocprofv3 trace
(gpurun_out/list_kernel_trace*.csv, optional): where does the issuing stream wait?
usage: python3 tools/list_progress.py [stride=10] [chain marker id=0: issuing stream; 1-4: D1-D4, 5: local-pair loss, 6: kNN]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, dev = 35, torch.device("cuda", 0)
torch.manual_seed(9999)
tr = PDGNTrainer(device=dev, distributed=False)
tr.train()
reals = synthetic_batch(B, dev)
g = torch.Generator().manual_seed(1234)
zs = [(noise(B, dev, g), noise(B, dev, g)) for _ in range(4)]
for i in range(3):
    tr.step(reals, *zs[i])
tr.capture_list(reals, *zs[0])
for i in range(5):
    tr.step_list(None, *zs[i % 4])
torch.cuda.synchronize()
L = _lib.lib()
label = int(sys.argv[2]) if len(sys.argv) > 2 else 0
main_chain = tr._list.labels.index(label)
n = 400
ms, pos = (ctypes.c_float * n)(), (ctypes.c_int * n)()
runs = []
for rep in range(3):
    torch.cuda.synchronize()
    k = L.pdgn_replay_probe_chain(tr._list._plan, main_chain, stride, ms, pos, n)
    runs.append([(pos[i], ms[i]) for i in range(k)])
best = min(runs, key=lambda r: r[-1][1])
print("chain %d: %d launches, %.2f ms from its first to its last (best of 3: %s)" % (label, best[-1][0], best[-1][1], ["%.2f" % r[-1][1] for r in runs]))
prev = (0, 0.0)
for p, t in best[1:]:
    print("  launches %4d .. %4d : %7.3f ms   (at %7.3f ms)" % (prev[0], p, t - prev[1], t))
    prev = (p, t)
sys.stdout.flush()
os._exit(0)          # (the kept hipGraph's teardown at interpreter exit faults on this ROCm; nothing left to do)
