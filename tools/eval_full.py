"""The evaluation the reference's README prices at "about 2 hours" (README.md:47; evaluation_metrics.py:85-121,172-200):
compute_all_metrics on N_sample x N_ref clouds of 2048 points -- three all-pairs matrices (sample-ref, ref-ref,
sample-sample), Chamfer + approximate EMD each -- on ONE MI355X, wall time and pairs/s.

    python tools/eval_full.py [S=1300] [R=1300] [out.json]

Synthetic clouds (uniform in the unit ball, shape_bbox-normalised like the test split): the kernels' work does not depend
on the shapes except through the EMD sweeps' zero runs (bench.py::eval_c5 measures those on the same distribution)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pdgn_amd import evaluation as ev

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1300
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1300
out_path = sys.argv[3] if len(sys.argv) > 3 else None
N = 2048
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(9999)


def clouds(n):
    x = torch.rand(n, N, 3, generator=g) * 2 - 1
    lo, hi = x.amin(dim=1, keepdim=True), x.amax(dim=1, keepdim=True)
    return ((x - (lo + hi) / 2) / (hi - lo).amax(dim=2, keepdim=True)).to(dev).contiguous()


smp, ref = clouds(S), clouds(R)
ev.pairwise_emd_cd(smp[:64], ref[:64])                       # warm-up: code objects, allocator
torch.cuda.synchronize()
times = {}
t_all = time.time()
for name, (a, b) in (("sample-ref", (smp, ref)), ("ref-ref", (ref, ref)), ("sample-sample", (smp, smp))):
    t0 = time.time()
    cd, emd = ev.pairwise_emd_cd(a, b)
    torch.cuda.synchronize()
    times[name] = time.time() - t0
    assert torch.isfinite(cd).all() and torch.isfinite(emd).all()
t_mats = time.time() - t_all
t0 = time.time()
res = ev.compute_all_metrics(smp, ref)
torch.cuda.synchronize()
t_metrics = time.time() - t0
pairs = S * R + R * R + S * S
line = {"what": "compute_all_metrics, %d x %d clouds of %d points (CD + approximate EMD, three all-pairs matrices), 1 MI355X" % (S, R, N),
        "reference": "README.md:47 'about 2 hours' (evaluation/evaluation_metrics.py:85-121,172-200)",
        "pairs": pairs, "seconds_three_matrices": round(t_mats, 3), "seconds_per_matrix": {k: round(v, 3) for k, v in times.items()},
        "pairs_per_s": round(pairs / t_mats, 1), "seconds_compute_all_metrics": round(t_metrics, 3),
        "metrics": {k: float(v) for k, v in res.items()}}
print(json.dumps(line))
if out_path:
    json.dump(line, open(out_path, "w"), indent=1)
