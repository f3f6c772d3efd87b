"""GPU parity of the fused local-pair loss kernels (csrc/localpair.hip) against the golden vectors
of the imported reference (utils/chamfer_loss.py, PDGNet_v2.compute_mean_covariance) and torch fp64."""
import numpy as np
import pytest
import torch

from oracle import cref, pdgnet_ref
from torch_standins import chamfer_min_torch, local_stats_torch

pytestmark = pytest.mark.gpu


def dev(a):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    return a.cuda()


def test_chamfer_loss_golden(golden):
    from pdgn_amd.losses import ChamferLoss
    g = golden("chamfer.npz")
    cl = ChamferLoss()
    np.testing.assert_allclose(cl(dev(g["a"]), dev(g["b"])).item(), g["chamfer_sum3"], rtol=1e-4)
    np.testing.assert_allclose(cl(dev(g["a9"]), dev(g["b9"])).item(), g["chamfer_sum9"], rtol=1e-4)


@pytest.mark.parametrize("B,M,N,D", [(3, 256, 256, 3), (2, 1024, 1024, 9), (2, 300, 77, 9), (2, 50, 1500, 3), (1, 40, 33, 5)])
def test_chamfer_gram_forward_backward(B, M, N, D):
    from pdgn_amd.losses import chamfer_min
    rng = np.random.default_rng(M + N + D)
    x = torch.from_numpy(rng.standard_normal((B, M, D)).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal((B, N, D)).astype(np.float32))
    gx = torch.from_numpy(rng.standard_normal((B, M)).astype(np.float32))
    gy = torch.from_numpy(rng.standard_normal((B, N)).astype(np.float32))
    xd, yd = dev(x).requires_grad_(True), dev(y).requires_grad_(True)
    minx, miny = chamfer_min(xd, yd)
    ((minx * dev(gx)).sum() + (miny * dev(gy)).sum()).backward()
    xr, yr = x.double().requires_grad_(True), y.double().requires_grad_(True)
    rx, ry = chamfer_min_torch(xr, yr)
    ((rx * gx.double()).sum() + (ry * gy.double()).sum()).backward()
    scale = float(rx.abs().max())
    np.testing.assert_allclose(minx.detach().cpu().numpy(), rx.detach().numpy(), rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(miny.detach().cpu().numpy(), ry.detach().numpy(), rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(yd.grad.cpu().numpy(), yr.grad.numpy(), rtol=1e-3, atol=1e-4)


def test_local_stats_golden(golden):
    from pdgn_amd.losses import local_stats
    g = golden("chamfer.npz")
    pts = g["mc_points"]                                     # (R,3,20) neighbourhoods
    R = pts.shape[0]
    xyz = np.ascontiguousarray(pts.transpose(0, 2, 1).reshape(1, R * 20, 3))
    idx = np.arange(R * 20, dtype=np.int32).reshape(1, R, 20)
    mu, cov = local_stats(dev(xyz), dev(idx))
    np.testing.assert_allclose(mu.cpu().numpy().reshape(R, 3, 1), g["mc_mu"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cov.cpu().numpy().reshape(R, 3, 3), g["mc_cov"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,N,M,K", [(2, 512, 256, 20), (3, 2048, 1024, 20), (1, 5000, 64, 7)])
def test_local_stats_forward_backward(B, N, M, K):
    from pdgn_amd.losses import local_stats
    rng = np.random.default_rng(N + M)
    xyz = torch.from_numpy(rng.standard_normal((B, N, 3)).astype(np.float32))
    idx = torch.from_numpy(rng.integers(0, N, (B, M, K)).astype(np.int32))
    gmu = torch.from_numpy(rng.standard_normal((B, M, 3)).astype(np.float32))
    gcov = torch.from_numpy(rng.standard_normal((B, M, 9)).astype(np.float32))
    xd = dev(xyz).requires_grad_(True)
    mu, cov = local_stats(xd, dev(idx))
    ((mu * dev(gmu)).sum() + (cov * dev(gcov)).sum()).backward()
    xr = xyz.double().requires_grad_(True)
    rmu, rcov = local_stats_torch(xr, idx)
    ((rmu * gmu.double()).sum() + (rcov * gcov.double()).sum()).backward()
    np.testing.assert_allclose(mu.detach().cpu().numpy(), rmu.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(cov.detach().cpu().numpy(), rcov.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=1e-4)


def test_local_pair_vs_oracle():
    """get_local_pair (:136-155) end to end: HIP kNN + local stats + Chamfer vs the oracle's
    C pointops + torch restatement, values and gradients."""
    from pdgn_amd.losses import LocalPairLoss
    rng = np.random.default_rng(5)
    p1 = torch.from_numpy(rng.standard_normal((2, 3, 256)).astype(np.float32))
    p2 = torch.from_numpy(rng.standard_normal((2, 3, 512)).astype(np.float32))
    a, b = dev(p1).requires_grad_(True), dev(p2).requires_grad_(True)
    mu, cov = LocalPairLoss(20)(a, b)
    (mu + cov).backward()
    ar, br = p1.clone().requires_grad_(True), p2.clone().requires_grad_(True)
    rmu, rcov = pdgnet_ref.local_pair(ar, br)
    (rmu + rcov).backward()
    np.testing.assert_allclose(mu.item(), rmu.item(), rtol=1e-4)
    np.testing.assert_allclose(cov.item(), rcov.item(), rtol=1e-4)
    np.testing.assert_allclose(a.grad.cpu().numpy(), ar.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(b.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("shape,target,scale", [((35, 1), 1.0, 0.5), ((35, 1), 0.0, 0.5), ((7, 3), 1.0, 1.0), ((5000,), -0.25, 2.0)])
def test_mse_against_a_constant_fwd_bwd(shape, target, scale):
    """losses.mse_const (pdgn_mse_const[_backward]) == scale * nn.MSELoss()(x, target): the adversarial terms of
    models/PDGNet_v2.py:186-190, 246-250, value and gradient, with a non-unit upstream gradient."""
    from pdgn_amd import losses
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(*shape, device="cuda", generator=g).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    out = losses.mse_const(x, target, scale)
    ref = torch.nn.functional.mse_loss(xr, torch.full_like(xr, target)) * scale
    np.testing.assert_allclose(out.item(), ref.item(), rtol=1e-6)
    (out * 1.7).backward()
    (ref * 1.7).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("b,m,n,d", [(3, 100, 70, 3), (2, 257, 512, 9), (35, 256, 128, 3), (35, 1024, 1024, 9)])
def test_chamfer_sum_fwd_bwd_vs_torch(b, m, n, d):
    """losses.chamfer_sum (one node: pdgn_chamfer_gram + pdgn_scaled_sum forward, pdgn_chamfer_gram_grad_uniform backward with
    the upstream scalar read on the device) against the torch stand-in, value and both gradients."""
    from pdgn_amd import losses
    from torch_standins import chamfer_sum_torch
    g = torch.Generator(device="cuda").manual_seed(b * m + n)
    x = torch.randn(b, m, d, device="cuda", generator=g).requires_grad_(True)
    y = torch.randn(b, n, d, device="cuda", generator=g).requires_grad_(True)
    xr, yr = x.detach().clone().requires_grad_(True), y.detach().clone().requires_grad_(True)
    out = losses.chamfer_sum(x, y, 0.25)
    ref = chamfer_sum_torch(xr, yr, 0.25)
    np.testing.assert_allclose(out.item(), ref.item(), rtol=2e-5)
    (out * 3.0).backward()
    (ref * 3.0).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y.grad.cpu().numpy(), yr.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
