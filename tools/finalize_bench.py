#!/usr/bin/env python3
"""pdgn_bn_stats_from_gemm_partials alone at the step's shapes: one launch (no scratch) against the two-launch form for long lists."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
L.pdgn_bn_blocks_scratch_doubles.restype = ctypes.c_longlong
for rows, C, block in [(35840, 512, 128), (35840, 512, 64), (71680, 256, 64), (179200, 256, 128), (358400, 64, 64), (358400, 512, 128), (358400, 512, 32), (179200, 256, 32), (17920, 1024, 64)]:
    nparts = -(-rows // block)
    part = torch.randn(nparts, 3 * C, device="cuda")
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    stats = torch.empty(4 * C, device="cuda")
    nd = L.pdgn_bn_blocks_scratch_doubles(C, ctypes.c_longlong(nparts))
    scr = torch.empty(max(nd, 1), dtype=torch.float64, device="cuda")
    res = []
    for use in (None, scr if nd > 0 else None):
        def run():
            L.pdgn_bn_stats_from_gemm_partials(ctypes.c_longlong(rows), C, ctypes.c_longlong(nparts), block, ctypes.c_float(1e-5),
                                               ctypes.c_float(0.1), ptr(g), ptr(b), None, ptr(rm), ptr(rv), ptr(part), ptr(stats), ptr(use),
                                               stream_of(part))
        for _ in range(5): run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200): run()
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 200 * 1e3)
    print("rows %7d C %5d block %3d nparts %5d: one launch %6.2f us, sliced (%d doubles of scratch) %6.2f us" % (rows, C, block, nparts, res[0], nd, res[1]))
