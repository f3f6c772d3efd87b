#!/usr/bin/env python3
"""Does torch's TunableOp (search over all rocBLAS / hipBLASLt solutions) beat the heuristic pick on the step's large GEMMs?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tn
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
# (form, M, N, K): nt = A(M,K) @ W(N,K)^T ; nn = A(M,K) @ B(K,N)
shapes = [("nt", 35840, 12832, 128), ("nt", 35840, 512, 5120), ("nn", 35840, 5120, 512), ("nn", 35840, 128, 12832),
          ("nt", 358400, 512, 64), ("nn", 358400, 64, 512), ("nt", 71680, 1024, 256), ("nn", 71680, 256, 1024),
          ("nt", 17920, 256, 2560), ("nt", 17920, 6432, 64), ("nn", 17920, 64, 6432), ("nt", 179200, 256, 64)]
if len(sys.argv) > 1 and sys.argv[1] == "tn":
    # weight gradients dW (N,K) = dY (M,N)^T @ X (M,K): the shapes pdgn_gemm_tn serves today
    shapes = [("tn", 35840, 512, 5120), ("tn", 35840, 12832, 128), ("tn", 71680, 1024, 256), ("tn", 358400, 512, 64),
              ("tn", 17920, 256, 2560), ("tn", 17920, 6432, 64), ("tn", 179200, 1024, 512), ("tn", 71680, 256, 128)]
tn.set_max_tuning_duration(10)          # ms per candidate
tn.set_max_tuning_iterations(10)
for form, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda")
    if form == "tn":
        A = torch.randn(M, N, device="cuda")           # dY
        B = torch.randn(M, K, device="cuda")           # X
        run = lambda: A.t().matmul(B)
    else:
        B = torch.randn(N, K, device="cuda") if form == "nt" else torch.randn(K, N, device="cuda")
        run = (lambda: torch.nn.functional.linear(A, B)) if form == "nt" else (lambda: A.matmul(B))
    tn.enable(False)
    base = {}
    for lib in (torch._C._BlasBackend.Cublaslt, torch._C._BlasBackend.Cublas):
        torch._C._set_blas_preferred_backend(lib)
        base[lib] = t(run)
    torch._C._set_blas_preferred_backend(torch._C._BlasBackend.Cublaslt)
    tn.enable(True); tn.tuning_enable(True)
    t0 = time.time(); run(); torch.cuda.synchronize(); tune_s = time.time() - t0
    tuned = t(run)
    fl = 2.0 * M * N * K
    b = min(base.values())
    print("%s M%-7d N%-6d K%-6d  hipblaslt %7.1f rocblas %7.1f | tuned %7.1f us (%.0f TF, %.2fx) tuning took %.1f s" % (
        form, M, N, K, base[torch._C._BlasBackend.Cublaslt], base[torch._C._BlasBackend.Cublas], tuned, fl / tuned / 1e6, b / tuned, tune_s), flush=True)

