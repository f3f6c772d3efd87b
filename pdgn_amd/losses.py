"""Losses of the PDGN training step: Chamfer (utils/chamfer_loss.py:13-38) and the
shape-preserving local-statistics loss (models/PDGNet_v2.py:127-155)."""
import torch
import torch.nn as nn

from . import pointops


class ChamferLoss(nn.Module):
    """utils/chamfer_loss.py:13-38: Gram-form P = |x|^2 + |y|^2 - 2<x,y> (no clamp) and the SUM
    (not mean) of the row minima and column minima over batch and points."""

    def forward(self, preds, gts):
        P = self.batch_pairwise_dist(gts, preds)
        return torch.min(P, 1)[0].sum() + torch.min(P, 2)[0].sum()

    @staticmethod
    def batch_pairwise_dist(x, y):
        zz = torch.bmm(x, y.transpose(2, 1))
        rx = (x * x).sum(dim=2, keepdim=True)                  # diag(x x^T), :29-35
        ry = (y * y).sum(dim=2).unsqueeze(1)
        return rx + ry - 2 * zz


def compute_mean_covariance(points):
    """models/PDGNet_v2.py:127-134.  points (R,3,k) -> mu (R,3,1), covariance (R,3,3)."""
    mu = points.mean(dim=-1, keepdim=True)
    tmp = points - mu
    return mu, torch.bmm(tmp, tmp.transpose(1, 2)) / points.shape[-1]


class LocalPairLoss(nn.Module):
    """get_local_pair :136-155: group both clouds around pt1's points (k = 20 nearest, HIP kNN +
    grouping), compare per-neighbourhood means and covariances with Chamfer, divide by M."""

    def __init__(self, nsample=20):
        super().__init__()
        self.nsample = nsample
        self.group = pointops.Gen_QueryAndGroupXYZ(radius=None, nsample=nsample, use_xyz=False)
        self.chamfer_loss = ChamferLoss()

    def forward(self, pt1, pt2):
        B, _, M = pt1.shape
        new_xyz = pt1.transpose(1, 2).contiguous()
        pt2_trans = pt2.transpose(1, 2).contiguous()
        g1 = self.group(new_xyz, new_xyz).transpose(1, 2).contiguous().view(-1, 3, self.nsample)
        g2 = self.group(pt2_trans, new_xyz).transpose(1, 2).contiguous().view(-1, 3, self.nsample)
        mu1, var1 = compute_mean_covariance(g1)
        mu2, var2 = compute_mean_covariance(g2)
        like_mu = self.chamfer_loss(mu1.view(B, -1, 3), mu2.view(B, -1, 3)) / float(M)
        like_var = self.chamfer_loss(var1.view(B, -1, 9), var2.view(B, -1, 9)) / float(M)
        return like_mu, like_var
