#!/usr/bin/env python3
"""Does the launch-list capture of the iteration with generator pass #1 on D4's stream (PDGN_PASS1_SIDE=1) survive when it runs in a
thread with a large stack?  (hipStreamEndCapture walks the captured graph recursively: on the default 8-MB stack the capture of two
concurrent generator passes ends in a segmentation fault.)  usage: p1_capture_probe.py [stack MB]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = 35
dev = torch.device("cuda", 0)
torch.manual_seed(9999)
tr = PDGNTrainer(device=dev)
tr.train()
reals = synthetic_batch(B, dev)
z = [(noise(B, dev), noise(B, dev)) for _ in range(8)]
for _ in range(2):
    tr.step(reals, *z[0])
torch.cuda.synchronize()
res = {}


def work():
    torch.cuda.set_device(dev)
    try:
        tr.capture_list(reals, *z[0])
        res["ok"] = True
    except Exception as e:
        res["err"] = repr(e)


threading.stack_size(mb << 20)
t = threading.Thread(target=work)
t.start()
t.join()
print("capture in a thread with %d MB of stack:" % mb, res, flush=True)
if res.get("ok"):
    print(tr._list.info, flush=True)
    for i in range(3):
        tr.step_list(None, *z[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        out = tr.step_list(None, *z[i % 8])
    torch.cuda.synchronize()
    print("ms/step", (time.perf_counter() - t0) / 20 * 1e3, {k: float(v) for k, v in out.items()}, flush=True)
