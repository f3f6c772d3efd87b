#!/usr/bin/env python3
"""Time the oracle's CPU G+D iteration on this host for several thread counts (sizing of cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cref, pdgnet_ref
from pdgn_amd.trainer import synthetic_batch
cref.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for th in [int(a) for a in sys.argv[2:]] or [16, 32, 64]:
    torch.set_num_threads(th)
    torch.manual_seed(0)
    tr = pdgnet_ref.TrainerRef()
    z = lambda b: torch.randn(b, 128) * 0.2
    reals = synthetic_batch(B, "cpu")
    t0 = time.perf_counter(); tr.step(reals, z(B), z(B)); t1 = time.perf_counter()
    tr.step(reals, z(B), z(B)); t2 = time.perf_counter()
    print("threads %3d  B=%d  first %.1f s  second %.1f s  -> %.0f points/s" % (th, B, t1 - t0, t2 - t1, B * 2048 / (t2 - t1)), flush=True)
