"""Launch-list replay vs the eager step: equality of results and time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch

B = int(os.environ.get("B", "35"))
dev = torch.device("cuda", 0)


def snapshot(tr):
    ts = []
    for net in [tr.G] + tr.D:
        ts += list(net.parameters()) + list(net.buffers())
    for opt in [tr.optG] + tr.optD:
        for st in opt.state.values():
            ts += [v for v in st.values() if torch.is_tensor(v)]
    return ts, [t.detach().clone() for t in ts]


def restore(snap):
    ts, vals = snap
    with torch.no_grad():
        for t, v in zip(ts, vals):
            t.copy_(v)


reals = synthetic_batch(B, dev)
gen = torch.Generator().manual_seed(5)
zs = [(noise(B, dev, gen), noise(B, dev, gen)) for _ in range(40)]
torch.manual_seed(9999)
tr = PDGNTrainer(device=dev, distributed=False)
tr.train()
for i in range(3):
    tr.step(reals, *zs[i])          # Adam state exists now
torch.cuda.synchronize()
snap = snapshot(tr)
eager = []
for i in range(3):
    o = tr.step(reals, *zs[10 + i])
    eager.append({k: v.item() for k, v in o.items()})
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    tr.step(reals, *zs[20 + i % 10])
torch.cuda.synchronize()
print("eager %.2f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
tr.capture_list(reals, *zs[0])
torch.cuda.synchronize()
print("capture_list %.1f s" % (time.perf_counter() - t0), tr._list.info)
print("chains (label, nodes):", list(zip(tr._list.labels, tr._list.sizes)))
restore(snap)
lst = []
for i in range(3):
    o = tr.step_list(reals, *zs[10 + i])
    lst.append({k: v.item() for k, v in o.items()})
for a, b in zip(eager, lst):
    print(" eager", {k: round(v, 6) for k, v in a.items()})
    print(" list ", {k: round(v, 6) for k, v in b.items()})
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(20):
        tr.step_list(None, *zs[20 + i % 10])
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("list %.2f ms/step (host issue %.2f ms/step)" % ((time.perf_counter() - t0) / 20 * 1e3, t_issue / 20 * 1e3))
t0 = time.perf_counter()
for i in range(20):
    tr.step(reals, *zs[20 + i % 10])
torch.cuda.synchronize()
print("eager again %.2f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
import ctypes
from pdgn_amd import _lib
us = (ctypes.c_double * 32)()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.lib().pdgn_replay_launch_timed(tr._list._plan, us)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("one list on an idle device: issue %.2f ms, done after %.2f ms; kernels %.2f memset %.2f waits %.2f records %.2f ms; per chain us: %s"
          % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, us[0] / 1e3, us[1] / 1e3, us[2] / 1e3, us[3] / 1e3, [round(us[4 + c]) for c in range(7)]))
