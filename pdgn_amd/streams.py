"""Side streams of the overlapped G+D schedule, chosen by the HARDWARE QUEUE they land on.

HIP multiplexes its streams onto a few hardware (AQL) queues -- 4 per process on this ROCm; 5 or more is ~50% SLOWER
on MI355X (GPU_MAX_HW_QUEUES sweep) -- and kernels of two streams that share a queue run back to
back however independent they are.  Which queue a stream gets depends on every stream created before it: with an RCCL
communicator alive (it creates its own streams) the same `torch.cuda.Stream()` calls that gave the default stream a
queue of its own put the feature-kNN stream and a discriminator stream on the default stream's queue, and the
iteration went from 38.0 to 41.4 ms with no collective running.  So the queues are not assumed, they are measured:
`pdgn_spin` (csrc/probe.hip) occupies a stream for a fixed wall time with one wavefront; two of them on two streams
finish in ~1x that time when the streams sit on different queues and ~2x when they share one.  The candidates are
grouped that way once per (device, issuing stream) and the roles are dealt out so that

  * the issuing (default) stream's queue carries nothing else,
  * the feature-kNN stream and the local-pair-loss stream share a queue (the kNN runs inside the generator passes, the
    loss after them),
  * D1 + D2 + D3 share one of the remaining queues and D4 has the other to itself: the backward of D4(G(z2)) is what the
    generator's backward waits for, and behind another discriminator's kernels in a shared queue it waited longer (round 4,
    launch-list step: 28.75 vs 29.0 ms; rounds 1-3 paired D1 + D3 / D2 + D4).

There is no reference counterpart: models/PDGNet_v2.py runs on one CUDA stream.
"""
import ctypes
import os
import time

import torch

from . import _lib

SPIN_US = 150
_CANDIDATES = 12
_PLANS = {}


def _spin(stream, us=SPIN_US):
    _lib.check(_lib.lib().pdgn_spin(ctypes.c_uint(us), ctypes.c_void_p(stream.cuda_stream)), "pdgn_spin")


def shares_queue(a, b, device, us=SPIN_US):
    """True when streams a and b execute one after the other (same hardware queue).  Best of three timings."""
    best = float("inf")
    for _ in range(3):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        _spin(a, us)
        _spin(b, us)
        a.synchronize()
        b.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best > 1.6 * us * 1e-6


def queue_classes(device, issuing, candidates):
    """Group `candidates` by hardware queue.  Returns (streams sharing the issuing stream's queue, [other groups])."""
    with_issuing, groups = [], []
    for c in candidates:
        if shares_queue(issuing, c, device):
            with_issuing.append(c)
            continue
        for g in groups:
            if shares_queue(g[0], c, device):
                g.append(c)
                break
        else:
            groups.append([c])
    return with_issuing, groups


class StreamPlan:
    """d[0..3]: the discriminators' streams; lp: local-pair loss; knn: feature kNN of the generator passes."""

    def __init__(self, d, lp, knn, n_queues, probed):
        self.d, self.lp, self.knn, self.n_queues, self.probed = d, lp, knn, n_queues, probed

    def __repr__(self):
        return "StreamPlan(queues besides the issuing stream's: %d, probed=%s)" % (self.n_queues, self.probed)


def plan(device):
    """The plan for work issued from torch's current stream on `device` (built once, then cached)."""
    device = torch.device(device)
    issuing = torch.cuda.current_stream(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), issuing.cuda_stream)
    got = _PLANS.get(key)
    if got is not None:
        return got
    capturing = torch.cuda.is_current_stream_capturing()
    cands = [torch.cuda.Stream(device=device) for _ in range(_CANDIDATES)]
    if capturing or os.environ.get("PDGN_STREAM_PROBE", "1") != "1":
        # no timing inside a graph capture (and an A/B switch): creation order, as before the probe existed
        p = _PLANS[key] = StreamPlan(cands[:4], cands[4], cands[5], 0, False)
        return p
    _, groups = queue_classes(device, issuing, cands)
    groups.sort(key=len, reverse=True)

    def take(g):                                    # a fresh stream of group g while it has one, else its first
        return g.pop() if len(g) > 1 else g[0]

    if len(groups) >= 3:
        # D1-D3 on one queue, D4 alone on another, local-pair loss + feature kNN on the third; PDGN_STREAM_LAYOUT
        # (six letters A-C for D1 D2 D3 D4 lp knn) is the A/B switch the layouts of DESIGN.md section 10b were tried with
        layout = os.environ.get("PDGN_STREAM_LAYOUT", "BBBCAA")
        by = dict(zip("ABC", groups[:3]))
        roles = [take(by[ch]) for ch in layout]
        d, lp, knn = roles[:4], roles[4], roles[5]
    elif len(groups) == 2:
        a, b = groups
        d = [take(a), take(b), take(a), take(b)]
        lp, knn = take(a), take(b)
    elif len(groups) == 1:
        a = groups[0]
        d = [take(a) for _ in range(4)]
        lp, knn = take(a), take(a)
    else:                                           # one hardware queue in all: overlap is impossible, order is kept
        d, lp, knn = cands[:4], cands[4], cands[5]
    p = _PLANS[key] = StreamPlan(d, lp, knn, len(groups), True)
    return p


def reset():
    _PLANS.clear()
