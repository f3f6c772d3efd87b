#!/usr/bin/env python3
"""Ablation timing of feat_knn_kernel: full vs MFMA-only vs selection-only (diagnostic builds)."""
import ctypes, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
csrc = os.path.join(R, "pdgn_amd", "csrc")
def build(tag, flags):
    so = "/tmp/fk_%s.so" % tag
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
                           "-ffp-contract=off", os.path.join(csrc, "feat_knn.hip"), "-o", so] + flags)
    return ctypes.CDLL(so)
libs = {"full": build("full", []), "no_select": build("nosel", ["-DFK_ABLATE_SELECT"]),
        "no_mfma": build("nomfma", ["-DFK_ABLATE_MFMA"])}
for (B, F, N) in [(35, 256, 1024), (35, 128, 512)]:
    x = torch.randn(B, F, N, device="cuda")
    idx = torch.empty(B, N, 10, device="cuda", dtype=torch.int32)
    sq = torch.empty(B, N, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for tag, L in libs.items():
        f = lambda: L.pdgn_feature_knn(B, F, N, 10, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(sq.data_ptr()), ctypes.c_void_p(idx.data_ptr()), st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        print("B%d F%d N%d %-10s %8.1f us" % (B, F, N, tag, s.elapsed_time(e) * 100))
