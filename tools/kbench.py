#!/usr/bin/env python3
"""Per-kernel timing of the pointops / structural-loss kernels at BASELINE.json sizes (GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdgn_amd import pointops as po
from pdgn_amd import structural_losses as sl


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    B = 35
    res = {}
    torch.manual_seed(0)
    for n, m in [(256, 256), (512, 256), (2048, 256), (1024, 512), (2048, 1024)]:
        xyz = torch.randn(B, n, 3, device="cuda")
        q = xyz[:, :m].contiguous()
        us = timeit(lambda: po.knnquery(20, xyz, q))
        byt = B * (12 * n + 12 * m + 8 * m * 20)
        res["knn_%d_%d" % (n, m)] = dict(us=round(us, 1), GBps=round(byt / us / 1e3, 1),
                                          Gdist_per_s=round(B * n * m / us / 1e3, 1))
        idx = po.knnquery(20, xyz, q)
        f = xyz.transpose(1, 2).contiguous()
        us = timeit(lambda: po.grouping(f, idx))
        byt = B * (4 * 3 * n + 4 * m * 20 + 4 * 3 * m * 20)
        res["group_fwd_%d_%d" % (n, m)] = dict(us=round(us, 1), GBps=round(byt / us / 1e3, 1))
        g = torch.randn(B, 3, m, 20, device="cuda")
        fr = f.clone().requires_grad_(True)
        out = po.grouping(fr, idx)
        us = timeit(lambda: torch.autograd.grad(out, fr, g, retain_graph=True))
        res["group_bwd_%d_%d" % (n, m)] = dict(us=round(us, 1), GBps=round(byt / us / 1e3, 1))
    for b in (64, 512):
        a = torch.rand(b, 2048, 3, device="cuda") * 2 - 1
        c = torch.rand(b, 2048, 3, device="cuda") * 2 - 1
        us = timeit(lambda: sl.nn_distance(a, c), iters=5, warm=1)
        res["nndist_b%d" % b] = dict(us=round(us, 1), Gdist_per_s=round(2 * b * 2048 * 2048 / us / 1e3, 1))
        us = timeit(lambda: sl.emd_cost(a, c), iters=2, warm=1)
        res["emd_fused_b%d" % b] = dict(us=round(us, 1), Gexp_per_s=round(27 * b * 2048 * 2048 / us / 1e3, 1))
    b = 64
    a = torch.rand(b, 2048, 3, device="cuda") * 2 - 1
    c = torch.rand(b, 2048, 3, device="cuda") * 2 - 1
    from pdgn_amd.structural_losses.match_cost import ApproxMatch, MatchCost
    us = timeit(lambda: ApproxMatch(a, c), iters=2, warm=1)
    res["approxmatch_b64"] = dict(us=round(us, 1), GBps=round(19 * b * 2048 * 2048 * 4 / us / 1e3, 1))
    mt, _ = ApproxMatch(a, c)
    us = timeit(lambda: MatchCost(a, c, mt), iters=3, warm=1)
    res["matchcost_b64"] = dict(us=round(us, 1), GBps=round(b * 2048 * 2048 * 4 / us / 1e3, 1))
    for k, v in res.items():
        print(k, json.dumps(v))


if __name__ == "__main__":
    main()
