"""Shared assertions on the BACKWARD of one composed G+D iteration (tests/golden/step_b8_graphs.npz).

The fixture holds, from the reference's own lossD.backward() / lossG.backward() (models/PDGNet_v2.py:189-256, composed
in tests/golden/gen_golden.py::gen_step): the norm of the generator's gradient, the gradient norm of EVERY generator
parameter, leading slices of thirteen of them and of three per discriminator, and the first Adam update of a weight
slice in units of the learning rate.  A trainer that has just run `step()` still holds that iteration's gradients in
`p.grad` (the discriminators are frozen during the generator update, so theirs are the D-update gradients)."""
import numpy as np
import torch

from hashweights import fill_module


def check_step_gradients(tr, g, rtol_norm=1e-2, slice_tol=2e-2):
    named = list(tr.G.named_parameters())
    names = [n for n, _ in named]
    assert names == [str(n) for n in g["g_param_names"]]
    gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in named)).item()
    np.testing.assert_allclose(gnorm, float(g["g_grad_norm"]), rtol=rtol_norm)
    norms = np.array([p.grad.norm().item() for _, p in named])
    ref_norms = g["g_param_grad_norms"]
    # per-parameter norms; biases in front of a training-mode BatchNorm have an identically zero gradient (the
    # reference holds rounding residue there, 1e-5..1e-7): those are compared on the scale of the whole gradient
    np.testing.assert_allclose(norms, ref_norms, rtol=rtol_norm, atol=1e-6 * float(g["g_grad_norm"]))
    sliced = [k[len("g_grad."):] for k in g.keys() if k.startswith("g_grad.")]
    assert len(sliced) >= 10
    params = dict(named)
    for n in sliced:
        want = g["g_grad." + n]
        got = params[n].grad.reshape(-1)[:want.size].cpu().numpy()
        scale = float(ref_norms[names.index(n)])
        err = np.abs(got - want).max()
        # elementwise: 2 % of the slice's largest element + 1e-3 of the parameter's gradient norm.  Measured spread of
        # two correct fp32 evaluations at this batch (BatchNorm over 8 samples amplifies rounding; torch-CPU stand-ins
        # against the reference): <= 0.9 % of the largest element; a wrong sign / wrong kernel is O(100 %)
        assert err <= slice_tol * np.abs(want).max() + 1e-3 * scale, (n, err, scale, np.abs(want).max())
    # -- discriminator backward (lossD.backward(), :189-222)
    for i, d in enumerate(tr.D, 1):
        dn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in d.parameters())).item()
        np.testing.assert_allclose(dn, float(g["d_grad_norm%d" % i]), rtol=rtol_norm, err_msg="D%d" % i)
        dp = dict(d.named_parameters())
        for k in [k for k in g.keys() if k.startswith("d%d_grad." % i)]:
            want = g[k]
            got = dp[k.split(".", 1)[1]].grad.reshape(-1)[:want.size].cpu().numpy()
            err = np.abs(got - want).max()
            assert err <= slice_tol * max(np.abs(want).max(), 1e-3 * float(g["d_grad_norm%d" % i])), (k, err)
    # -- first Adam step in units of lr: -g / (|g| + eps): O(1) and sign-sensitive (the updated weights themselves
    # move by 1e-4 against |w| = 0.05: a relative tolerance on them cannot fail)
    w0 = fill_module(type(tr.G)(), salt=1).fc1[0].weight.detach()[:4, :8].numpy()
    upd = (tr.G.fc1[0].weight.detach()[:4, :8].cpu().numpy() - w0) / 1e-4
    ref_upd = g["g_fc1_update_over_lr"]
    big = np.abs(ref_upd) > 0.5                                  # |g| well above Adam's eps: the update is -sign(g)
    assert big.sum() >= 8 and (np.sign(upd[big]) == np.sign(ref_upd[big])).all()
    np.testing.assert_allclose(upd, ref_upd, atol=0.05)


def assert_block_gradients(mod, g, x, pc, bound=5e-6):
    """Backward of a deconvolution block against the imported reference's fixture, on the scale of each tensor:
    max |ours - reference| <= bound * max |reference| for grad_x, grad_pc and every grad.<param>.
    Measured (tools/backward_error.py, profiles/r05_backward_error.txt): 1.7e-7 .. 2.3e-6 on all four fixtures for the torch
    stand-ins (fp32), both bf16 matrix instructions and the fp32 matrix instructions -- and THE SAME 1.7e-7 .. 1.3e-6 for the
    host logic evaluated in fp64: what is left is the reference's own fp32 rounding, not the re-association.  bound = 2x the worst
    measured.  (Rounds 2-4 compared elementwise at rtol 1e-3: that decade came from elements near zero, where a relative error
    of two fp32 evaluations means nothing.)  Biases in front of a training-mode BatchNorm have an identically zero gradient:
    the reference holds ~1e-9 of rounding residue there, this code exact zeros -- compared against the block's largest weight
    gradient (measured residue 5e-8 .. 1.9e-7 of it)."""
    import numpy as np

    def rel(a, b):
        return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
    assert rel(x.grad.cpu().numpy(), g["grad_x"]) <= bound, ("grad_x", rel(x.grad.cpu().numpy(), g["grad_x"]))
    if pc is not None:
        assert rel(pc.grad.cpu().numpy(), g["grad_pc"]) <= bound, ("grad_pc", rel(pc.grad.cpu().numpy(), g["grad_pc"]))
    gmax = max(np.abs(g["grad." + n]).max() for n, _ in mod.named_parameters())
    for n, p in mod.named_parameters():
        ref, got = g["grad." + n], p.grad.cpu().numpy()
        if np.abs(ref).max() < 1e-6 * gmax:                     # analytically zero (see above)
            assert np.abs(got.astype(np.float64) - ref).max() <= 1e-6 * gmax, n
        else:
            assert rel(got, ref) <= bound, (n, rel(got, ref))
