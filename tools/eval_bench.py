"""Time the all-pairs evaluation (evaluation_metrics.py:85-121) on the pair-list kernels vs the
reference's expand-per-sample loop run on our batched kernels.  2048-point clouds as in test()."""
import sys, time
import torch
sys.path.insert(0, ".")
from pdgn_amd import evaluation as ev
from pdgn_amd.losses import chamfer_min
from pdgn_amd.structural_losses import emd_cost

dev = torch.device("cuda:0")
S, R, N = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 128, 2048
torch.manual_seed(0)
smp = torch.randn(S, N, 3, device=dev)
ref = torch.randn(R, N, 3, device=dev)


def loop():
    cds, emds = [], []
    for i in range(S):
        a = smp[i:i + 1].expand(R, -1, -1).contiguous()
        minx, miny = chamfer_min(a, ref)
        cds.append(miny.mean(1) + minx.mean(1))
        emds.append(emd_cost(a, ref) / N)
    return torch.stack(cds), torch.stack(emds)


for name, fn in (("pair-list", lambda: ev.pairwise_emd_cd(smp, ref)), ("expand-loop", loop)):
    fn(); torch.cuda.synchronize()
    t = time.time(); out = fn(); torch.cuda.synchronize(); dt = time.time() - t
    print("%-12s S=%d R=%d N=%d: %.3f s  (%.0f pairs/s)" % (name, S, R, N, dt, S * R / dt))
