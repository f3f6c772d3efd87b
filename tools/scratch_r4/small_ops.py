import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda", 0)
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B = 35
for (Fc, Mw) in ((32, 3232), (64, 6432), (128, 12832)):
    c = torch.randn(B, Fc, device=dev); W = torch.randn(Mw, Fc, device=dev); g = torch.randn(B, Mw, device=dev)
    print("Yc = const W^T  (35 x %d x %d): fwd %.1f us, dconst %.1f us, dW %.1f us  (W = %.1f MB)" % (
        Fc, Mw, timeit(lambda: torch.nn.functional.linear(c, W)), timeit(lambda: g.matmul(W)), timeit(lambda: g.t().matmul(c)), Mw * Fc * 4 / 1e6))
g = torch.randn(B, 512, device=dev); W0 = torch.randn(256, 768, device=dev); b0 = torch.randn(256, device=dev); drb = torch.randn(B, 256, device=dev)
print("head: addmm %.1f us, drb^T g %.1f us, drb W0 %.1f us" % (timeit(lambda: torch.addmm(b0, g, W0[:, :512].t())), timeit(lambda: drb.t().mm(g)), timeit(lambda: drb.mm(W0[:, :512]))))
for (N, K) in ((512, 1024), (256, 512), (4096, 128)):
    dp = torch.randn(B, N, device=dev); W = torch.randn(N, K, device=dev)
    print("small layer dx = dpre W (35 x %d x %d): %.1f us" % (N, K, timeit(lambda: dp.matmul(W))))
x = torch.randn(B, 2048, 256, device=dev)
print("sum over points (35, 2048, 256) -> (35, 256): %.1f us;  rows (71680, 64).sum(0): %.1f us" % (timeit(lambda: x.sum(dim=1)), timeit(lambda: x.view(-1, 64)[:71680].sum(dim=0))))
y = torch.randn(B, 1024, 512, device=dev)
print("permute copy (35,1024,256,2)->(35,2,1024,256): %.1f us" % timeit(lambda: y.view(B, 1024, 256, 2).permute(0, 3, 1, 2).contiguous()))
a = torch.randn(B, 256, 1024, device=dev); cst = torch.randn(B, 128, device=dev)
print("kNN input: transpose+cat+contiguous (35, 128+128, 1024): %.1f us" % timeit(lambda: torch.cat((cst.unsqueeze(2).expand(-1, -1, 1024), a[:, :128]), 1).contiguous()))
