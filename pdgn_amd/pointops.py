"""Drop-in for the reference's ``lib/pointops/functions/pointops.py`` on MI355X.

Same callables, argument order, dtypes and autograd contract as the reference
(file:line cited per symbol); the native side is libpdgn_hip.so (include/pdgn_hip.h)
instead of the ``pointops_cuda`` pybind module.  Differences, all deliberate:
  * outputs are allocated on the INPUT's device (the reference uses the legacy
    ``torch.cuda.IntTensor(...)`` constructors = current device, pointops.py:425-426);
  * kernels run on torch's current stream (the reference's grouping/interpolation use the
    legacy null stream, grouping_cuda_kernel.cu:85);
  * launch failures raise ``PdgnHipError`` (the reference calls ``exit(-1)``).
There is no CPU path: CPU tensors raise.
"""
import ctypes
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn
from ._fn import Function

from . import _lib
from ._lib import check, ptr, require, stream_of

F32, I32 = torch.float32, torch.int32


# ------------------------------------------------------------------ hot-path Functions
class KNNQuery(Function):
    """pointops.py:408-434.  (nsample, xyz (b,n,3), new_xyz (b,m,3)|None) -> idx (b,m,nsample) int32."""

    @staticmethod
    def forward(ctx, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor = None) -> torch.Tensor:
        if new_xyz is None:
            new_xyz = xyz
        require(xyz, "xyz", F32, 3)
        require(new_xyz, "new_xyz", F32, 3)
        b, m, _ = new_xyz.size()
        n = xyz.size(1)
        idx = torch.empty((b, m, nsample), dtype=I32, device=xyz.device)
        dist2 = torch.empty((b, m, nsample), dtype=F32, device=xyz.device)
        check(_lib.lib().pdgn_knnquery(b, n, m, int(nsample), ptr(xyz), ptr(new_xyz), ptr(idx),
                                       ptr(dist2), stream_of(xyz)), "pdgn_knnquery")
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


knnquery = KNNQuery.apply


def knnquery_with_dist(nsample, xyz, new_xyz=None):
    """Like ``knnquery`` but also returns the squared distances the kernel computes
    (the reference writes and then drops them, pointops.py:426-428)."""
    if new_xyz is None:
        new_xyz = xyz
    require(xyz, "xyz", F32, 3)
    require(new_xyz, "new_xyz", F32, 3)
    b, m, _ = new_xyz.size()
    idx = torch.empty((b, m, nsample), dtype=I32, device=xyz.device)
    dist2 = torch.empty((b, m, nsample), dtype=F32, device=xyz.device)
    check(_lib.lib().pdgn_knnquery(b, xyz.size(1), m, int(nsample), ptr(xyz), ptr(new_xyz), ptr(idx),
                                   ptr(dist2), stream_of(xyz)), "pdgn_knnquery")
    return idx, dist2


class Grouping(Function):
    """pointops.py:122-151.  features (b,c,n), idx (b,m,nsample) -> (b,c,m,nsample);
    backward = scatter-add into (b,c,n)."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        require(features, "features", F32, 3)
        require(idx, "idx", I32, 3)
        b, c, n = features.size()
        _, m, nsample = idx.size()
        output = torch.empty((b, c, m, nsample), dtype=F32, device=features.device)
        check(_lib.lib().pdgn_grouping_forward(b, c, n, m, nsample, ptr(features), ptr(idx), ptr(output),
                                               stream_of(features)), "pdgn_grouping_forward")
        ctx.for_backwards = (idx, n)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        idx, n = ctx.for_backwards
        b, c, m, nsample = grad_out.size()
        grad_features = torch.zeros((b, c, n), dtype=F32, device=grad_out.device)
        grad_out_data = grad_out.data.contiguous()
        check(_lib.lib().pdgn_grouping_backward(b, c, n, m, nsample, ptr(grad_out_data), ptr(idx),
                                                ptr(grad_features), stream_of(grad_out_data)),
              "pdgn_grouping_backward")
        return grad_features, None


grouping = Grouping.apply


class NearestNeighbor(Function):
    """pointops.py:61-83.  unknown (b,n,3), known (b,m,3) -> (dist (b,n,3) = sqrt(dist2), idx (b,n,3))."""

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        require(unknown, "unknown", F32, 3)
        require(known, "known", F32, 3)
        b, n, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty((b, n, 3), dtype=F32, device=unknown.device)
        idx = torch.empty((b, n, 3), dtype=I32, device=unknown.device)
        check(_lib.lib().pdgn_nearestneighbor(b, n, m, ptr(unknown), ptr(known), ptr(dist2), ptr(idx),
                                              stream_of(unknown)), "pdgn_nearestneighbor")
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


nearestneighbor = NearestNeighbor.apply


class Interpolation(Function):
    """pointops.py:86-119.  features (b,c,m), idx/weight (b,n,3) -> (b,c,n); grad wrt features."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        require(features, "features", F32, 3)
        require(idx, "idx", I32, 3)
        require(weight, "weight", F32, 3)
        b, c, m = features.size()
        n = idx.size(1)
        ctx.interpolation_for_backward = (idx, weight, m)
        output = torch.empty((b, c, n), dtype=F32, device=features.device)
        check(_lib.lib().pdgn_interpolation_forward(b, c, m, n, ptr(features), ptr(idx), ptr(weight),
                                                    ptr(output), stream_of(features)),
              "pdgn_interpolation_forward")
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.interpolation_for_backward
        b, c, n = grad_out.size()
        grad_features = torch.zeros((b, c, m), dtype=F32, device=grad_out.device)
        grad_out_data = grad_out.data.contiguous()
        check(_lib.lib().pdgn_interpolation_backward(b, c, n, m, ptr(grad_out_data), ptr(idx), ptr(weight),
                                                     ptr(grad_features), stream_of(grad_out_data)),
              "pdgn_interpolation_backward")
        return grad_features, None, None


interpolation = Interpolation.apply


# ------------------------------------------------------------------ pure-torch members of the API
def pairwise_distances(x, y=None):
    """pointops.py:346-365."""
    x_norm = (x ** 2).sum(1).view(-1, 1)
    if y is not None:
        y_t, y_norm = y.transpose(0, 1), (y ** 2).sum(1).view(1, -1)
    else:
        y_t, y_norm = x.transpose(0, 1), x_norm.view(1, -1)
    return torch.clamp(x_norm + y_norm - 2.0 * torch.mm(x, y_t), 0.0, np.inf)


def _naive_sorted_idx(xyz, new_xyz):
    if new_xyz is None:
        new_xyz = xyz
    dist = (new_xyz.unsqueeze(2) - xyz.unsqueeze(1)).pow(2).sum(dim=3)
    return torch.sort(dist, dim=2)[1]


def knnquery_naive(nsample, xyz, new_xyz=None):
    """pointops.py:368-405 (torch sort of the full distance matrix)."""
    return _naive_sorted_idx(xyz, new_xyz)[:, :, 0:nsample].int()


def knnquery_exclude(nsample, xyz, new_xyz=None):
    """pointops.py:437-474: ranks 1..nsample."""
    return _naive_sorted_idx(xyz, new_xyz)[:, :, 1:nsample + 1].int()


class Gathering(Function):
    """pointops.py:33-58.  features (b,c,n), idx (b,m) int32 -> (b,c,m); backward = scatter-add into (b,c,n)."""

    @staticmethod
    def forward(ctx, features, idx):
        require(features, "features", F32, 3)
        require(idx, "idx", I32, 2)
        b, c, n = features.shape
        m = idx.shape[1]
        out = torch.empty((b, c, m), dtype=F32, device=features.device)
        check(_lib.lib().pdgn_gathering_forward(b, c, n, m, ptr(features), ptr(idx), ptr(out), stream_of(features)),
              "pdgn_gathering_forward")
        ctx.for_backwards = (idx, c, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, c, n = ctx.for_backwards
        b, m = idx.shape
        grad_out = grad_out.contiguous()
        grad = torch.zeros((b, c, n), dtype=F32, device=grad_out.device)
        check(_lib.lib().pdgn_gathering_backward(b, c, n, m, ptr(grad_out), ptr(idx), ptr(grad), stream_of(grad_out)),
              "pdgn_gathering_backward")
        return grad, None


gathering = Gathering.apply
featuregather = Gathering.apply            # pointops.py:225-260: same contraction (max_feature (b,c,n), idx (b,m))


def grouping_int(features, idx):
    """pointops.py:154-173: int64 features (b,c,n), idx (b,m,ns) int32 -> (b,c,m,ns) int64."""
    require(features, "features", torch.int64, 3)
    require(idx, "idx", I32, 3)
    b, c, n = features.shape
    _, m, ns = idx.shape
    out = torch.empty((b, c, m, ns), dtype=torch.int64, device=features.device)
    check(_lib.lib().pdgn_grouping_int_forward(b, c, n, m, ns, ptr(features), ptr(idx), ptr(out), stream_of(features)),
          "pdgn_grouping_int_forward")
    return out


def ballquery(radius, nsample, xyz, new_xyz):
    """pointops.py:176-198 / ballquery_cuda_kernel.cu:47-80: the first <= nsample points with d2 < r^2 in index
    order, padded with the first hit (idx 0 when the ball is empty: the output starts zeroed, :190)."""
    require(xyz, "xyz", F32, 3)
    require(new_xyz, "new_xyz", F32, 3)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.zeros((b, m, nsample), dtype=I32, device=xyz.device)
    check(_lib.lib().pdgn_ballquery(b, n, m, ctypes.c_float(radius), int(nsample), ptr(new_xyz), ptr(xyz), ptr(idx),
                                    stream_of(xyz)), "pdgn_ballquery")
    return idx


def furthestsampling(xyz, m):
    """pointops.py:12-30 / sampling_cuda_kernel.cu:59-168: iterative farthest point sampling from point 0."""
    require(xyz, "xyz", F32, 3)
    b, n, _ = xyz.shape
    idx = torch.zeros((b, m), dtype=I32, device=xyz.device)
    temp = torch.full((b, n), 1e10, dtype=F32, device=xyz.device)
    check(_lib.lib().pdgn_furthestsampling(b, n, int(m), ptr(xyz), ptr(temp), ptr(idx), stream_of(xyz)),
          "pdgn_furthestsampling")
    return idx


def featuredistribute(max_xyz, xyz):
    """pointops.py:201-222: index of the nearest max_xyz (b,n,3) point for every xyz (b,m,3) point."""
    require(max_xyz, "max_xyz", F32, 3)
    require(xyz, "xyz", F32, 3)
    b, n, _ = max_xyz.shape
    m = xyz.shape[1]
    out = torch.empty((b, m), dtype=I32, device=xyz.device)
    check(_lib.lib().pdgn_featuredistribute(b, n, m, ptr(max_xyz), ptr(xyz), ptr(out), stream_of(xyz)),
          "pdgn_featuredistribute")
    return out


def labelstat_idx(nsample, label_stat, idx):
    """pointops.py:291-315: new_label_stat[b,j,:] = sum_s label_stat[b, idx[b,j,s], :] (int32)."""
    require(label_stat, "label_stat", I32, 3)
    require(idx, "idx", I32, 3)
    b, n, nclass = label_stat.shape
    m = idx.shape[1]
    out = torch.empty((b, m, nclass), dtype=I32, device=idx.device)
    check(_lib.lib().pdgn_labelstat_idx(b, n, m, int(nsample), nclass, ptr(label_stat), ptr(idx), ptr(out),
                                        stream_of(idx)), "pdgn_labelstat_idx")
    return out


def labelstat_ballrange(radius, xyz, new_xyz, label_stat):
    """pointops.py:263-288: label histogram over all points with d2 < r^2."""
    require(xyz, "xyz", F32, 3)
    require(new_xyz, "new_xyz", F32, 3)
    require(label_stat, "label_stat", I32, 3)
    b, n, nclass = label_stat.shape
    m = new_xyz.shape[1]
    out = torch.empty((b, m, nclass), dtype=I32, device=xyz.device)
    check(_lib.lib().pdgn_labelstat_ballrange(b, n, m, ctypes.c_float(radius), nclass, ptr(new_xyz), ptr(xyz),
                                              ptr(label_stat), ptr(out), stream_of(xyz)), "pdgn_labelstat_ballrange")
    return out


def labelstat_and_ballquery(radius, nsample, xyz, new_xyz, label_stat):
    """pointops.py:318-343 -> (new_label_stat (b,m,nclass), idx (b,m,nsample)): ball query whose kept hits are also
    summed into the label histogram."""
    require(xyz, "xyz", F32, 3)
    require(new_xyz, "new_xyz", F32, 3)
    require(label_stat, "label_stat", I32, 3)
    b, n, nclass = label_stat.shape
    m = new_xyz.shape[1]
    out = torch.empty((b, m, nclass), dtype=I32, device=xyz.device)
    idx = torch.zeros((b, m, nsample), dtype=I32, device=xyz.device)
    check(_lib.lib().pdgn_labelstat_and_ballquery(b, n, m, ctypes.c_float(radius), int(nsample), nclass, ptr(new_xyz),
                                                  ptr(xyz), ptr(label_stat), ptr(idx), ptr(out), stream_of(xyz)),
          "pdgn_labelstat_and_ballquery")
    return out, idx


# ------------------------------------------------------------------ grouping Modules
class _QueryBase(nn.Module):
    def __init__(self, radius=None, nsample=32, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def _query(self, xyz, new_xyz, nsample=None):
        nsample = self.nsample if nsample is None else nsample
        if self.radius is not None:
            return ballquery(self.radius, nsample, xyz, new_xyz)
        return knnquery(nsample, xyz, new_xyz)


class Gen_QueryAndGroupXYZ(_QueryBase):
    """pointops.py:670-703 -- the one PDGNet_v2 uses (models/PDGNet_v2.py:115):
    xyz (b,n,3), new_xyz (b,m,3) -> grouped xyz (b,3,m,nsample)."""

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor = None) -> torch.Tensor:
        if new_xyz is None:
            new_xyz = xyz
        idx = self._query(xyz, new_xyz)
        return grouping(xyz.transpose(1, 2).contiguous(), idx)


class QueryAndGroup(_QueryBase):
    """pointops.py:525-566."""
    _dilate = 1

    def forward(self, xyz, new_xyz=None, features=None, idx=None):
        if new_xyz is None:
            new_xyz = xyz
        if idx is None:
            idx = self._query(xyz, new_xyz, self._dilate * self.nsample)
            if self._dilate > 1:                              # QueryAndGroup_Dilate :593-596
                pick = np.random.permutation(self._dilate * self.nsample)[:self.nsample]
                idx = idx[:, :, torch.as_tensor(pick, device=idx.device)].contiguous()
        grouped_xyz = grouping(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = grouping(features, idx)
            if self.use_xyz:
                return torch.cat([grouped_xyz, grouped_features], dim=1)
            return grouped_features
        assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        return grouped_xyz


class QueryAndGroup_Dilate(QueryAndGroup):
    """pointops.py:568-615: query 2*nsample neighbours, keep a random half."""
    _dilate = 2


class Le_QueryAndGroup(_QueryBase):
    """pointops.py:617-668: returns (grouped_xyz - centre, grouped features)."""

    def forward(self, xyz, new_xyz=None, features=None, idx=None):
        if new_xyz is None:
            new_xyz = xyz
        if idx is None:
            idx = self._query(xyz, new_xyz)
        grouped_xyz = grouping(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            return grouped_xyz, grouping(features, idx)
        assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        return grouped_xyz, grouped_xyz


class Le_QueryAndGroup_SameSize(Le_QueryAndGroup):
    """pointops.py:476-522 (the reference dereferences new_xyz before its None check, :497-499;
    here the check comes first)."""

    def forward(self, xyz, new_xyz=None, features=None, idx=None):
        if new_xyz is None:
            new_xyz = xyz
        assert xyz.size() == new_xyz.size()
        return super().forward(xyz, new_xyz, features, idx)


class Le_QueryAndGroup_OnlyFeature(_QueryBase):
    """pointops.py:705-751."""

    def forward(self, xyz, new_xyz=None, features=None, idx=None):
        if new_xyz is None:
            new_xyz = xyz
        if idx is None:
            idx = self._query(xyz, new_xyz)
        assert features is not None, "Le_QueryAndGroup_OnlyFeature needs features"
        return grouping(features, idx)


class GroupAll(nn.Module):
    """pointops.py:753-777."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            if self.use_xyz:
                return torch.cat([grouped_xyz, grouped_features], dim=1)
            return grouped_features
        return grouped_xyz
