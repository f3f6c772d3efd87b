#!/usr/bin/env python3
"""(rows, C) of every fused BatchNorm call in one G+D step, with counts and per-call device time (torch profiler)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch

L = _lib.lib()
calls = collections.Counter()


class Proxy:
    def __getattr__(self, name):
        f = getattr(L, name)
        if name not in ("pdgn_bn_stats", "pdgn_bn_act_forward", "pdgn_bn_act_backward", "pdgn_bn_act_maxpool",
                        "pdgn_bn_act_maxpool_backward", "pdgn_bn_eval_stats"):
            return f

        def wrapped(*a):
            key = tuple(getattr(x, "value", x) for x in a[:3] if isinstance(x, int) or hasattr(x, "value"))
            calls[(name, key)] += 1
            return f(*a)
        return wrapped


B = int(sys.argv[1]) if len(sys.argv) > 1 else 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
_lib._lib = Proxy()
tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
for (name, key), cnt in sorted(calls.items(), key=lambda kv: (kv[0][0], -(kv[0][1][0] or 0) * (kv[0][1][1] if len(kv[0][1]) > 1 and kv[0][1][1] else 1))):
    print("%-30s %-28s x%d" % (name, key, cnt))
