#!/usr/bin/env python3
"""pdgn_gemm_nt (every tile configuration) vs torch's library GEMM on the step's forward / input-gradient shapes:
correctness against fp64 and time.  usage: nt_bench.py [big|small|all] [--check-only]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
L = _lib.lib()
L.pdgn_gemm_nt_stat_rows.restype = ctypes.c_longlong


def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


def nt(A, W, C, bias=None, add=None, st=None):
    M, K = A.shape
    N = W.shape[0]
    rc = L.pdgn_gemm_nt(ctypes.c_longlong(M), N, K, ptr(A), A.stride(0), ptr(W), W.stride(0), ptr(bias), ptr(add),
                        add.stride(0) if add is not None else 0, ptr(C), C.stride(0), ptr(st), stream_of(A))
    assert rc == 0, rc


BIG = [(35840, 12832, 128), (35840, 512, 5120), (35840, 128, 12832), (35840, 5120, 512), (358400, 512, 64), (358400, 64, 512),
       (71680, 1024, 256), (71680, 256, 1024), (71680, 256, 128), (17920, 6432, 64), (17920, 256, 2560), (71680, 256, 256),
       (17920, 64, 6432), (17920, 2560, 256)]
SMALL = [(4480, 1600, 32), (4480, 64, 640), (8960, 256, 32), (8960, 128, 64), (8960, 3232, 32), (89600, 64, 16), (89600, 128, 64),
         (8960, 128, 1280), (17920, 256, 64), (17920, 512, 256), (179200, 64, 16), (179200, 256, 64), (35840, 256, 128),
         (35840, 512, 256), (71680, 128, 64), (8960, 64, 256), (71680, 64, 256), (358400, 64, 16), (1000, 36, 20), (130, 260, 4)]
which = sys.argv[1] if len(sys.argv) > 1 else "big"
shapes = {"big": BIG, "small": SMALL, "all": BIG + SMALL}[which]
check_only = "--check-only" in sys.argv
cfgs = [int(c) for c in os.environ.get("NT_CFGS", "0,1,2,3").split(",")]
torch.manual_seed(0)
TUNED = None
if os.environ.get("NT_TUNED", "1") == "1" and os.path.exists("tools/tunableop_gfx950_r01.csv"):
    import torch.cuda.tunable as TUNED
    TUNED.enable(True); TUNED.tuning_enable(False)
    if not TUNED.read_file("tools/tunableop_gfx950_r01.csv"):
        TUNED = None
    else:
        TUNED.enable(False)
for M, N, K in shapes:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    bias = torch.randn(N, device="cuda"); add = torch.randn(M, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    ref = torch.nn.functional.linear(A, W)
    if M * N * K <= 2e10:
        ref = (A.double() @ W.double().t()).float()
    scale = ref.abs().max().item()
    fl = 2.0 * M * N * K
    line = "M%-7d N%-6d K%-6d" % (M, N, K)
    u_lib = t(lambda: torch.nn.functional.linear(A, W)) if not check_only else 0.0
    line += " | lib %7.1f us %6.1f TF" % (u_lib, fl / max(u_lib, 1e-9) / 1e6)
    if TUNED is not None and not check_only:            # the best rocBLAS / hipBLASLt solution for this exact problem
        TUNED.enable(True)
        u_t = t(lambda: torch.nn.functional.linear(A, W))
        TUNED.enable(False)
        line += " | tuned %7.1f us %6.1f TF" % (u_t, fl / u_t / 1e6)
    for cfg in cfgs:
        os.environ["PDGN_NT_CFG"] = str(cfg)
        C.fill_(float("nan"))
        nt(A, W, C)
        err = ((C - ref).abs().max() / scale).item()
        # epilogues: bias + addend + statistics
        rows = L.pdgn_gemm_nt_stat_rows(ctypes.c_longlong(M), N, K)
        st = torch.full((rows, 2 * N), float("nan"), device="cuda")
        C2 = torch.empty_like(C)
        nt(A, W, C2, bias, add, st)
        ref2 = ref + bias + add
        err2 = ((C2 - ref2).abs().max() / scale).item()
        s = st.double().sum(0)
        es = ((s[:N] - C2.double().sum(0)).abs().max() / C2.double().abs().sum(0).max()).item()
        eq = ((s[N:] - (C2.double() ** 2).sum(0)).abs().max() / (C2.double() ** 2).sum(0).max()).item()
        bad = "" if max(err, err2, es, eq) < 2e-5 else "  <-- BAD"
        if check_only:
            line += " | c%d err %.1e/%.1e st %.1e/%.1e%s" % (cfg, err, err2, es, eq, bad)
        else:
            u = t(lambda: nt(A, W, C))
            u2 = t(lambda: nt(A, W, C2, None, None, st))
            line += " | c%d %7.1f us %6.1f TF (+st %7.1f) e %.0e%s" % (cfg, u, fl / u / 1e6, u2, max(err, err2, es, eq), bad)
    print(line, flush=True)
    del A, W, C, add, ref
