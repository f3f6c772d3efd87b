"""Launch-list replay of a captured iteration (csrc/replay.hip).

`torch.cuda.CUDAGraph(keep_graph=True)` records the iteration once (stream capture: every kernel's launch parameters and
every cross-stream dependency); the recorded graph is never instantiated or launched.  `LaunchList` reads it back and
re-issues its launches with plain HIP calls on the streams the eager schedule uses -- no autograd nodes, allocations or
ctypes marshalling per launch, and none of hipGraphLaunch's own scheduling (measured slower than the eager step on ROCm 7.2,
DESIGN.md section 10b).  There is no reference counterpart: models/PDGNet_v2.py issues its iteration op by op from Python.
"""
import ctypes

import torch

from . import _lib
from ._lib import check

MAIN, D0, LP, KNN = 0, 1, 5, 6            # marker ids of the overlapped schedule's streams (D1..D4 = 1..4)
POINT_BASE = 64                           # marker ids from here on: host points (csrc/replay.hip)


class Recorder:
    """Host points of a capture: what the iteration does that a stream capture cannot record -- the RCCL all-reduces of the
    data-parallel step.  While a Recorder is active (`with recorder:` around the capture), `point(fn)` launches a marker on
    the current stream instead of calling `fn`; the launch list reports the marker's position and chain, and the replay calls
    `fn` there, on that chain's stream, between two ranges of the list (PDGNTrainer.step_list)."""

    active = None

    def __init__(self):
        self.fns = []

    def __enter__(self):
        Recorder.active = self
        return self

    def __exit__(self, *exc):
        Recorder.active = None

    def point(self, fn):
        pid = POINT_BASE + len(self.fns)
        self.fns.append(fn)
        mark(pid, torch.cuda.current_stream())
        return pid


def recorder():
    return Recorder.active


def mark(stream_id, stream):
    """Tag `stream` inside a capture: the chain of nodes that follows on it is replayed on the stream given for this id."""
    check(_lib.lib().pdgn_replay_marker(int(stream_id), ctypes.c_void_p(stream.cuda_stream)), "pdgn_replay_marker")


class LaunchList:
    """The launches of one captured graph.  `graph` must have been created with keep_graph=True and stay alive as long
    as this object (its memory pool holds every buffer the launches address)."""

    def __init__(self, graph):
        self.graph = graph
        self._plan = ctypes.c_void_p()
        L = _lib.lib()
        raw = graph.raw_cuda_graph()
        rc = L.pdgn_replay_build(ctypes.c_void_p(raw), ctypes.byref(self._plan))
        if rc != 0:
            raise _lib.PdgnHipError("pdgn_replay_build failed with %d (a node type or copy shape the launch list does not re-issue)" % rc)
        counts = (ctypes.c_int * 8)()
        check(L.pdgn_replay_info(self._plan, counts), "pdgn_replay_info")
        self.info = dict(zip(("nodes", "kernels", "memsets", "memcpys", "empties", "chains", "events", "labelled"), counts))
        nc = self.info["chains"]
        labels, sizes = (ctypes.c_int * nc)(), (ctypes.c_int * nc)()
        check(L.pdgn_replay_chains(self._plan, labels, sizes), "pdgn_replay_chains")
        self.labels, self.sizes = list(labels), list(sizes)
        self._bound = None
        L.pdgn_replay_points.restype = ctypes.c_int
        n = L.pdgn_replay_points(self._plan, None, None, None, 0)
        if n < 0:
            raise _lib.PdgnHipError("pdgn_replay_points failed with %d" % n)
        ids, pos, chain = (ctypes.c_int * max(n, 1))(), (ctypes.c_int * max(n, 1))(), (ctypes.c_int * max(n, 1))()
        L.pdgn_replay_points(self._plan, ids, pos, chain, n)
        self.points = [(pos[i], ids[i] - POINT_BASE, self.labels[chain[i]]) for i in range(n)]     # (list position, point index, chain's marker id)
        # the last launch of the issuing stream's chain is behind the last launch of every other chain (a join that no later launch of
        # the issuing stream follows would be dropped, and the next iteration could overtake a side stream's tail)
        self.joined = bool(L.pdgn_replay_joined(self._plan))

    def bind(self, streams, spare):
        """streams: {marker id: torch stream}; spare: streams for chains without a marker (dealt out round-robin)."""
        key = tuple(sorted((k, s.cuda_stream) for k, s in streams.items())) + tuple(s.cuda_stream for s in spare)
        if key == self._bound:
            return
        L = _lib.lib()
        nxt = 0
        for c, lab in enumerate(self.labels):
            if lab in streams:
                s = streams[lab]
            else:
                s = spare[nxt % len(spare)]
                nxt += 1
            check(L.pdgn_replay_set_stream(self._plan, c, ctypes.c_void_p(s.cuda_stream)), "pdgn_replay_set_stream")
        self._keep = (dict(streams), list(spare))
        self._bound = key

    def launch(self, lo=0, hi=None):
        """Issue nodes [lo, hi) of the list (all of it by default)."""
        check(_lib.lib().pdgn_replay_launch_range(self._plan, int(lo), self.info["nodes"] if hi is None else int(hi)),
              "pdgn_replay_launch_range")

    def kernel_nodes(self, sym, grid_x=-1):
        """List positions of the kernel nodes launched through host symbol `sym` (an int address) with grid.x == grid_x."""
        L = _lib.lib()
        n = L.pdgn_replay_kernel_nodes(self._plan, ctypes.c_void_p(sym), int(grid_x), None, 0)
        if n <= 0:
            return []
        pos = (ctypes.c_int * n)()
        L.pdgn_replay_kernel_nodes(self._plan, ctypes.c_void_p(sym), int(grid_x), pos, n)
        return list(pos)

    def neighbor(self, pos, direction):
        """(list position, kind, kernel symbol) of the next / previous node of `pos`'s chain; position -1 when there is none."""
        kind, sym = ctypes.c_int(-1), ctypes.c_void_p()
        L = _lib.lib()
        L.pdgn_replay_chain_neighbor.restype = ctypes.c_int
        p = L.pdgn_replay_chain_neighbor(self._plan, int(pos), int(direction), ctypes.byref(kind), ctypes.byref(sym))
        return p, kind.value, sym.value

    def time_spans(self, spans, slots):
        """Bracket spans [(first, last), ...] of the list (each inside one chain) with timing events on that chain's stream for the
        next `slots` passes (csrc/replay.hip); [] switches the timing off."""
        n = len(spans)
        first = (ctypes.c_int * max(n, 1))(*[a for a, _ in spans])
        last = (ctypes.c_int * max(n, 1))(*[b for _, b in spans])
        check(_lib.lib().pdgn_replay_time_spans(self._plan, first, last, n, int(slots)), "pdgn_replay_time_spans")
        self._timed = (n, int(slots))

    def timed_ms(self):
        """[[ms, ...] per timed span]: durations of the passes recorded since time_spans (waits for them)."""
        n, slots = getattr(self, "_timed", (0, 0))
        if n == 0:
            return []
        ms, cnt = (ctypes.c_float * (n * slots))(), (ctypes.c_int * n)()
        check(_lib.lib().pdgn_replay_time_read(self._plan, ms, cnt), "pdgn_replay_time_read")
        return [[ms[i * slots + j] for j in range(cnt[i])] for i in range(n)]

    def position(self, label, fraction):
        """List position right after `fraction` of the launches of the chain with marker id `label`."""
        c = self.labels.index(label)
        nth = max(0, min(self.sizes[c] - 1, int(self.sizes[c] * fraction)))
        pos = _lib.lib().pdgn_replay_position(self._plan, c, nth)
        return pos + 1 if pos >= 0 else self.info["nodes"]

    def __del__(self):
        try:
            if self._plan:
                _lib.lib().pdgn_replay_destroy(self._plan)
                self._plan = ctypes.c_void_p()
        except Exception:
            pass
