"""How far is ApproxMatch's `match` from the C oracle's, element by element?"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cref
from pdgn_amd import structural_losses as sl
from pdgn_amd.structural_losses.match_cost import ApproxMatch
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for b, n, m in [(3, 256, 256), (2, 512, 512), (2, 300, 100), (2, 100, 250), (1, 2048, 2048)]:
    rng = np.random.default_rng(n * 3 + m)
    a = rng.uniform(-1, 1, (b, n, 3)).astype(np.float32)
    c = rng.uniform(-1, 1, (b, m, 3)).astype(np.float32)
    ref = cref.approxmatch(a, c)
    got = ApproxMatch(dev(a), dev(c))[0].cpu().numpy()
    err = np.abs(got - ref)
    print("shape", (b, n, m), "max ref %.3g" % ref.max(), "max abs err %.3g" % err.max())
    for thr in (1e-3, 1e-4, 1e-5, 1e-6, 1e-8, 1e-12):
        sel = ref > thr
        if sel.any():
            print("   ref > %.0e: %9d elems, max rel err %.3g" % (thr, sel.sum(), (err[sel] / ref[sel]).max()))
    for atol in (2e-5, 1e-6, 1e-7, 1e-8):
        print("   rtol 1e-4 + atol %.0e: violations %d" % (atol, (err > 1e-4 * np.abs(ref) + atol).sum()))
rng = np.random.default_rng(2)
a = rng.uniform(-1, 1, (2, 200, 3)).astype(np.float32)
c = rng.uniform(-1, 1, (2, 200, 3)).astype(np.float32)
ta, tc = dev(a).requires_grad_(True), dev(c).requires_grad_(True)
sl.match_cost(ta, tc).sum().backward()
match = cref.approxmatch(a, c)
g1, g2 = cref.matchcost_grad(a, c, match)
for got, ref in ((ta.grad.cpu().numpy(), g1), (tc.grad.cpu().numpy(), g2)):
    err = np.abs(got - ref)
    print("grad: max |ref| %.3g max abs err %.3g; rtol1e-4+atol1e-6 violations %d, atol 1e-5: %d" % (
        np.abs(ref).max(), err.max(), (err > 1e-4 * np.abs(ref) + 1e-6).sum(), (err > 1e-4 * np.abs(ref) + 1e-5).sum()))
