#!/usr/bin/env python3
"""pdgn_bn_stats_from_gemm_partials (cl_finalize_blocks_kernel) alone at the step's shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
for rows, C, block in [(35840, 512, 128), (35840, 512, 64), (71680, 256, 64), (179200, 256, 128), (358400, 64, 64), (17920, 1024, 64)]:
    nparts = -(-rows // block)
    part = torch.randn(nparts, 3 * C, device="cuda")
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    stats = torch.empty(4 * C, device="cuda")
    def run():
        L.pdgn_bn_stats_from_gemm_partials(ctypes.c_longlong(rows), C, ctypes.c_longlong(nparts), block, ctypes.c_float(1e-5),
                                           ctypes.c_float(0.1), ptr(g), ptr(b), None, ptr(rm), ptr(rv), ptr(part), ptr(stats), stream_of(part))
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200): run()
    e.record(); torch.cuda.synchronize()
    print("rows %7d C %5d block %3d nparts %5d: %6.2f us per launch (back to back)" % (rows, C, block, nparts, s.elapsed_time(e) / 200 * 1e3))
