#!/usr/bin/env python3
"""Where the default stream's time goes inside the OVERLAPPED step (HIP events on the default stream, no tracer: a
kernel tracer serialises the queues and stretches the step from 38 to 57 ms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B, STEPS = 35, 20
tr = PDGNTrainer(device="cuda"); tr.train()
if os.environ.get("PDGN_X_NOLP") == "1":       # diagnosis only: how much of the step is waiting for the local-pair loss?
    tr.similar_terms = lambda clouds, pairs: {p: (clouds[0].sum() * 0, clouds[0].sum() * 0) for p in pairs}
reals = synthetic_batch(B, "cuda")
zs = [(noise(B, "cuda"), noise(B, "cuda")) for _ in range(STEPS + 5)]
for i in range(5):
    tr.step(reals, *zs[i])
torch.cuda.synchronize()
runs = []
for i in range(STEPS):
    ev = []
    def mark(name, ev=ev):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append((name, e))
    st = tr._state(reals, *zs[5 + i])
    st["mark"] = mark
    tr._step_overlapped(None, None, None, st=st)
    runs.append(ev)
torch.cuda.synchronize()
names = [n for n, _ in runs[0]][1:]
tot = 0.0
print("default stream, mean over %d steps (the backward's own D(gen)/loss adjoints run on their forward streams)" % STEPS)
for k, n in enumerate(names):
    ms = sum(r[k][1].elapsed_time(r[k + 1][1]) for r in runs) / STEPS
    tot += ms
    print("  %-40s %7.2f ms" % (n, ms))
print("  %-40s %7.2f ms" % ("sum (event to event, one step)", tot))
if os.environ.get("PDGN_X_PERSTEP") == "1":
    for k, n in enumerate(names):
        print("  per step %-28s" % n, " ".join("%.1f" % r[k][1].elapsed_time(r[k + 1][1]) for r in runs))
gap = sum(runs[i][-1][1].elapsed_time(runs[i + 1][0][1]) for i in range(STEPS - 1)) / (STEPS - 1)
print("  %-40s %7.2f ms" % ("between steps", gap))
