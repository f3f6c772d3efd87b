#!/usr/bin/env python3
"""Host-side wall time of one G+D step by autograd Function (forward / backward) and by C-ABI entry point: where do the
~34 ms of host issue time go?"""
import collections, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib, deconv, fused, losses
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch

acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(cls, name):
    for meth in ("forward", "backward"):
        f = getattr(cls, meth)
        def make(f, key):
            def g(*a, **k):
                t0 = time.perf_counter()
                try:
                    return f(*a, **k)
                finally:
                    e = acc[key]; e[0] += 1; e[1] += time.perf_counter() - t0
            return staticmethod(g)
        setattr(cls, meth, make(f, "%s.%s" % (name, meth)))
for mod in (fused, deconv, losses):
    for name in dir(mod):
        obj = getattr(mod, name)
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            wrap(obj, name)
# C ABI calls
L = _lib.lib()
cacc = collections.defaultdict(lambda: [0, 0.0])
class Proxy:
    def __getattr__(self, n):
        fn = getattr(L, n)
        def g(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            e = cacc[n]; e[0] += 1; e[1] += time.perf_counter() - t0
            return r
        return g
_lib._LIB = None
orig_lib = _lib.lib
_lib.lib = lambda: Proxy()
for m in (fused, deconv, losses):
    if hasattr(m, "_lib"):
        pass
B = 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
zs = [(noise(B, "cuda"), noise(B, "cuda")) for _ in range(20)]
for i in range(5):
    tr.step(reals, *zs[i])
torch.cuda.synchronize()
acc.clear(); cacc.clear()
N = 10
t0 = time.perf_counter()
for i in range(N):
    tr.step(reals, *zs[5 + i])
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host issue %.2f ms/step (instrumented)" % ((t1 - t0) / N * 1e3))
print("-- autograd Functions (host wall ms/step, calls/step)")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %-44s %7.3f ms  %6.1f calls  %6.1f us/call" % (k, t / N * 1e3, n / N, t / n * 1e6))
print("  total in Functions: %.2f ms/step" % (sum(t for _, t in acc.values()) / N * 1e3))
print("-- C ABI calls (ctypes, ms/step)")
for k, (n, t) in sorted(cacc.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-44s %7.3f ms  %6.1f calls  %6.1f us/call" % (k, t / N * 1e3, n / N, t / n * 1e6))
print("  total in C ABI calls: %.2f ms/step, %d calls/step" % (sum(t for _, t in cacc.values()) / N * 1e3, sum(n for n, _ in cacc.values()) / N))
