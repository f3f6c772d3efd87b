"""Pre-split second operand (pdgn_split_bf16x3 + pdgn_gemm_nt_ps) against the unsplit entry points: bit-identity and time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of, check
L = _lib.lib()
dev = torch.device("cuda", 0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def split(w, want_t):
    n, k = w.shape
    ldp = (k + 7) // 8 * 8
    P = torch.zeros((3, n, ldp), dtype=torch.int16, device=dev)
    ldt = (n + 7) // 8 * 8
    PT = torch.zeros((3, k, ldt), dtype=torch.int16, device=dev) if want_t else None
    check(L.pdgn_split_bf16x3(n, k, ptr(w), w.stride(0), ptr(P), ldp, ctypes.c_longlong(n * ldp), ptr(PT), ldt,
                              ctypes.c_longlong(k * ldt if want_t else 0), stream_of(w)), "split")
    return P, PT


for (m, n, k) in [(35840, 512, 5120), (35840, 12832, 128), (17920, 256, 2560), (71680, 1024, 256), (358400, 512, 64), (1000, 132, 260)]:
    torch.manual_seed(m + n + k)
    a = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.1
    b = torch.randn(n, device=dev)
    P, PT = split(w, True)
    c0, c1 = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
    nt = lambda: check(L.pdgn_gemm_nt(ctypes.c_longlong(m), n, k, ptr(a), k, ptr(w), k, ptr(b), None, 0, ptr(c0), n, None, stream_of(a)), "nt")
    ps = lambda: check(L.pdgn_gemm_nt_ps(ctypes.c_longlong(m), n, k, ptr(a), k, ptr(P), P.shape[2], ctypes.c_longlong(P.shape[1] * P.shape[2]), P.shape[0],
                                         ptr(b), None, 0, ptr(c1), n, None, None, 0, 1, 0, None, 0, stream_of(a)), "ps")
    nt(); ps()
    same = torch.equal(c0, c1)
    t0, t1 = timeit(nt), timeit(ps)
    msg = "nt  m %6d n %5d k %5d  identical %s  unsplit %7.1f us  pre-split %7.1f us (%.3fx, %.0f TF)" % (m, n, k, same, t0, t1, t1 / t0, 2.0 * m * n * k / t1 / 1e6)
    # input gradient dX = dY W through the planes of W^T
    dy = torch.randn(m, n, device=dev)
    d0, d1 = torch.empty(m, k, device=dev), torch.empty(m, k, device=dev)
    nn = lambda: check(L.pdgn_gemm_nn(ctypes.c_longlong(m), k, n, ptr(dy), n, ptr(w), k, None, None, 0, ptr(d0), k, None, stream_of(a)), "nn")
    pst = lambda: check(L.pdgn_gemm_nt_ps(ctypes.c_longlong(m), k, n, ptr(dy), n, ptr(PT), PT.shape[2], ctypes.c_longlong(PT.shape[1] * PT.shape[2]), PT.shape[0],
                                          None, None, 0, ptr(d1), k, None, None, 0, 1, 0, None, 0, stream_of(a)), "pst")
    if k % 4 == 0 and n % 4 == 0:
        nn(); pst()
        err = ((d0 - d1).abs().max() / d0.abs().max()).item()
        t2, t3 = timeit(nn), timeit(pst)
        msg += " | nn unsplit %7.1f us  W^T planes %7.1f us (%.3fx)  rel diff %.1e" % (t2, t3, t3 / t2, err)
    print(msg, flush=True)
    del a, w, c0, c1, dy, d0, d1
tsp = None
w = torch.randn(12832, 128, device=dev)
print("split 12832 x 128: both orientations %.1f us, planes only %.1f us" % (timeit(lambda: split(w, True)), timeit(lambda: split(w, False))))
