import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from pdgn_amd import _lib
from pdgn_amd._lib import ptr, stream_of
_lib.lib()
L = ctypes.CDLL("tools/micro/bin/libfk_timing.so")
B, k, F, N = 35, 10, 256, 1024
x = torch.nn.functional.leaky_relu(torch.randn(B, F, N, device="cuda"))
idx = torch.empty(B, N, k, device="cuda", dtype=torch.int32); sq = torch.empty(B, N, device="cuda")
for _ in range(3):
    L.pdgn_feature_knn(B, F, N, k, ptr(x), ptr(sq), ptr(idx), stream_of(x))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
print("rc", L.fk_dbg_copy(buf))
t = np.array(buf, dtype=np.int64).reshape(8, 64)
t0 = t[0, 0]
names = ["P start", "P mfma done", "P epi done", "C start", "C select done", "C merge done", "after barrier", "C flush done"]
for it in range(12):
    print("it %2d " % it + "  ".join("%s %7d" % (names[s], t[s, it] - t0) for s in (0,1,2,3,4,7,5,6)))

b2 = (ctypes.c_ulonglong * 16)()
L.fk_dbg2_copy(b2)
v = list(b2)
print("flush phases (cycles):", [int(v[i + 1] - v[i]) for i in range(5)], " [append+load, bisect, compact, rank, final]")
print("select phases (cycles):", [int(v[i + 1] - v[i]) for i in range(6, 10)], " [pass A, bisect, tau, pass B]")
