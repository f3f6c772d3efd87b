"""Losses of the PDGN training step: Chamfer (utils/chamfer_loss.py:13-38) and the
shape-preserving local-statistics loss (models/PDGNet_v2.py:127-155), on fused HIP kernels
(csrc/localpair.hip): no (B,M,N) distance matrix, no (B,3,M,20) grouped tensor, no bmm."""
import ctypes

import torch
import torch.nn as nn
from ._fn import Function

from . import _lib, pointops
from ._lib import check, ptr, require, stream_of
from .fused import _zeros

F32, I32 = torch.float32, torch.int32


class ChamferGram(Function):
    """x (B,M,D), y (B,N,D) -> (min_j P[i,j] (B,M), min_i P[i,j] (B,N)) with the reference's
    Gram-form P = |x_i|^2 + |y_j|^2 - 2<x_i,y_j> (chamfer_loss.py:22-38)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = x.contiguous(), y.contiguous()
        require(x, "x", F32, 3)
        require(y, "y", F32, 3)
        b, m, d = x.shape
        n = y.shape[1]
        minx = torch.empty((b, m), dtype=F32, device=x.device)
        miny = torch.empty((b, n), dtype=F32, device=x.device)
        argx = torch.empty((b, m), dtype=I32, device=x.device)
        argy = torch.empty((b, n), dtype=I32, device=x.device)
        check(_lib.lib().pdgn_chamfer_gram(b, m, n, d, ptr(x), ptr(y), ptr(minx), ptr(argx), ptr(miny), ptr(argy),
                                           stream_of(x)), "pdgn_chamfer_gram")
        ctx.save_for_backward(x, y, argx, argy)
        ctx.mark_non_differentiable(argx, argy)
        return minx, miny

    @staticmethod
    def backward(ctx, gminx, gminy):
        x, y, argx, argy = ctx.saved_tensors
        b, m, d = x.shape
        n = y.shape[1]
        gx, gy = torch.empty_like(x), torch.empty_like(y)
        gminx, gminy = gminx.contiguous(), gminy.contiguous()
        check(_lib.lib().pdgn_chamfer_gram_grad(b, m, n, d, ptr(x), ptr(y), ptr(gminx), ptr(argx), ptr(gminy),
                                                ptr(argy), ptr(gx), ptr(gy), stream_of(x)),
              "pdgn_chamfer_gram_grad")
        return gx, gy


def chamfer_min(x, y):
    return ChamferGram.apply(x, y)


class ChamferSum(Function):
    """scale * (sum_i min_j P + sum_j min_i P) as ONE autograd node: the minima of both directions land in one
    buffer (one reduction), and the adjoint feeds the uniform upstream gradient straight to the scatter kernel --
    the per-term sum / add / div / expand launches of the 12 local-pair terms disappear."""

    @staticmethod
    def forward(ctx, x, y, scale):
        x, y = x.contiguous(), y.contiguous()
        require(x, "x", F32, 3)
        require(y, "y", F32, 3)
        b, m, d = x.shape
        n = y.shape[1]
        mins = torch.empty((b * (m + n),), dtype=F32, device=x.device)
        args = torch.empty((b * (m + n),), dtype=I32, device=x.device)
        minx, miny, argx, argy = mins[:b * m], mins[b * m:], args[:b * m], args[b * m:]
        check(_lib.lib().pdgn_chamfer_gram(b, m, n, d, ptr(x), ptr(y), ptr(minx), ptr(argx), ptr(miny), ptr(argy),
                                           stream_of(x)), "pdgn_chamfer_gram")
        ctx.save_for_backward(x, y, args)
        ctx.scale = float(scale)
        out = torch.empty((), dtype=F32, device=x.device)
        check(_lib.lib().pdgn_scaled_sum(ctypes.c_longlong(b * (m + n)), ptr(mins), ctypes.c_float(ctx.scale), ptr(out),
                                         stream_of(x)), "pdgn_scaled_sum")
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, args = ctx.saved_tensors
        b, m, d = x.shape
        n = y.shape[1]
        gbuf = torch.empty((b * (m + n) * d,), dtype=F32, device=x.device)      # gx | gy: one zero-fill inside the call
        gx, gy = gbuf[:b * m * d].view(b, m, d), gbuf[b * m * d:].view(b, n, d)
        g = g.contiguous()
        check(_lib.lib().pdgn_chamfer_gram_grad_uniform(b, m, n, d, ptr(x), ptr(y), ptr(g), ctypes.c_float(ctx.scale),
                                                        ptr(args[:b * m]), ptr(args[b * m:]), ptr(gx), ptr(gy), stream_of(x)),
              "pdgn_chamfer_gram_grad_uniform")
        return gx, gy, None


class MseConst(Function):
    """scale * nn.MSELoss()(x, target * ones_like(x)) -- the adversarial terms mse(D(x), 1) / mse(D(x), 0) with their 1/2
    (models/PDGNet_v2.py:186-190, 246-250) -- as one launch forward and one backward (csrc/loss_small.hip)."""

    @staticmethod
    def forward(ctx, x, target, scale):
        x = x.contiguous()
        require(x, "x", F32)
        out = torch.empty((), dtype=F32, device=x.device)
        check(_lib.lib().pdgn_mse_const(ctypes.c_longlong(x.numel()), ptr(x), ctypes.c_float(target), ctypes.c_float(scale),
                                        ptr(out), stream_of(x)), "pdgn_mse_const")
        ctx.save_for_backward(x)
        ctx.cfg = (float(target), float(scale))
        return out

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        target, scale = ctx.cfg
        dx = torch.empty_like(x)
        g = g.contiguous()
        check(_lib.lib().pdgn_mse_const_backward(ctypes.c_longlong(x.numel()), ptr(x), ctypes.c_float(target),
                                                 ctypes.c_float(scale), ptr(g), ptr(dx), stream_of(x)),
              "pdgn_mse_const_backward")
        return dx, None, None


def mse_const(x, target, scale=1.0):
    """scale * mean((x - target)^2) (MseConst)."""
    return MseConst.apply(x, float(target), float(scale))


def chamfer_sum(x, y, scale=1.0):
    """scale * ChamferLoss-style sum of both directions' minima (utils/chamfer_loss.py:16-20)."""
    return ChamferSum.apply(x, y, scale)


class LocalStats(Function):
    """xyz (B,N,3), idx (B,M,K) int32 -> mu (B,M,3), cov (B,M,9): grouping (:142-145) +
    compute_mean_covariance (:127-134) in one pass; backward scatters onto xyz."""

    @staticmethod
    def forward(ctx, xyz, idx):
        require(xyz, "xyz", F32, 3)
        require(idx, "idx", I32, 3)
        b, n, _ = xyz.shape
        _, m, k = idx.shape
        mu = torch.empty((b, m, 3), dtype=F32, device=xyz.device)
        cov = torch.empty((b, m, 9), dtype=F32, device=xyz.device)
        check(_lib.lib().pdgn_local_stats(b, n, m, k, ptr(xyz), ptr(idx), ptr(mu), ptr(cov), stream_of(xyz)),
              "pdgn_local_stats")
        ctx.save_for_backward(xyz, idx)
        return mu, cov

    @staticmethod
    def backward(ctx, dmu, dcov):
        xyz, idx = ctx.saved_tensors
        b, n, _ = xyz.shape
        _, m, k = idx.shape
        dxyz = _zeros(tuple(xyz.shape), xyz.device)           # (a slice of the backward pass's zero arena: no fill launch)
        dmu, dcov = dmu.contiguous(), dcov.contiguous()
        check(_lib.lib().pdgn_local_stats_backward(b, n, m, k, ptr(xyz), ptr(idx), ptr(dmu), ptr(dcov), ptr(dxyz),
                                                   stream_of(xyz)), "pdgn_local_stats_backward")
        return dxyz, None


def local_stats(xyz, idx):
    return LocalStats.apply(xyz, idx)


def knnquery(nsample, xyz, new_xyz):
    return pointops.knnquery(nsample, xyz, new_xyz)


class ChamferLoss(nn.Module):
    """utils/chamfer_loss.py:13-38: SUM (not mean) over batch and points of the row minima and
    the column minima of the Gram-form pairwise matrix."""

    def forward(self, preds, gts):
        return chamfer_sum(gts, preds, 1.0)


def compute_mean_covariance(points):
    """models/PDGNet_v2.py:127-134 (torch ops; kept for API parity).  points (R,3,k)."""
    mu = points.mean(dim=-1, keepdim=True)
    tmp = points - mu
    return mu, torch.bmm(tmp, tmp.transpose(1, 2)) / points.shape[-1]


class LocalPairLoss(nn.Module):
    """get_local_pair :136-155: neighbourhood (k = 20) means and covariances of both clouds around
    pt1's points, compared with Chamfer, divided by M."""

    def __init__(self, nsample=20):
        super().__init__()
        self.nsample = nsample
        self.chamfer_loss = ChamferLoss()

    def stats(self, cloud_cl, query_cl):
        """cloud (B,N,3), queries (B,M,3) point-major -> (mu (B,M,3), cov (B,M,9))."""
        idx = knnquery(self.nsample, cloud_cl, query_cl)
        return local_stats(cloud_cl, idx)

    def forward(self, pt1, pt2, self_stats=None):
        """pt1 (B,3,M), pt2 (B,3,N>=M).  `self_stats` = stats(pt1, pt1) when the caller already has
        them (pt1 is grouped around itself identically in every pair it heads, :232-237)."""
        M = pt1.shape[2]
        new_xyz = pt1.transpose(1, 2).contiguous()
        mu1, var1 = self_stats if self_stats is not None else self.stats(new_xyz, new_xyz)
        mu2, var2 = self.stats(pt2.transpose(1, 2).contiguous(), new_xyz)
        return chamfer_sum(mu2, mu1, 1.0 / float(M)), chamfer_sum(var2, var1, 1.0 / float(M))
