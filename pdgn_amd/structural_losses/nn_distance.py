"""Chamfer nearest-neighbour distance.  Mirrors
evaluation/pytorch_structural_losses/nn_distance.py:7-41 (NNDistanceFunction) and the pybind
entry points NNDistance / NNDistanceGrad (src/structural_loss.cpp:80-124)."""
import torch
from .._fn import Function

from .. import _lib
from .._lib import check, ptr, require, stream_of

F32, I32 = torch.float32, torch.int32


def NNDistance(set_d, set_q):
    """structural_loss.cpp:80-100 -> [dist1 (b,n), idx1 (b,n) i32, dist2 (b,m), idx2 (b,m) i32]."""
    require(set_d, "set_d", F32, 3)
    require(set_q, "set_q", F32, 3)
    b, n, _ = set_d.shape
    m = set_q.shape[1]
    dev = set_d.device
    dist1 = torch.empty((b, n), dtype=F32, device=dev)
    idx1 = torch.empty((b, n), dtype=I32, device=dev)
    dist2 = torch.empty((b, m), dtype=F32, device=dev)
    idx2 = torch.empty((b, m), dtype=I32, device=dev)
    check(_lib.lib().pdgn_nndistance(b, n, ptr(set_d), m, ptr(set_q), ptr(dist1), ptr(idx1), ptr(dist2),
                                     ptr(idx2), stream_of(set_d)), "pdgn_nndistance")
    return [dist1, idx1, dist2, idx2]


def NNDistanceGrad(set_d, set_q, idx1, idx2, grad_dist1, grad_dist2):
    """structural_loss.cpp:102-124 -> [grad1 (b,n,3), grad2 (b,m,3)]."""
    b, n, _ = set_d.shape
    m = set_q.shape[1]
    grad_dist1 = grad_dist1.contiguous()
    grad_dist2 = grad_dist2.contiguous()
    grad1 = torch.empty((b, n, 3), dtype=F32, device=set_d.device)
    grad2 = torch.empty((b, m, 3), dtype=F32, device=set_d.device)
    check(_lib.lib().pdgn_nndistance_grad(b, n, ptr(set_d), m, ptr(set_q), ptr(grad_dist1), ptr(idx1),
                                          ptr(grad_dist2), ptr(idx2), ptr(grad1), ptr(grad2),
                                          stream_of(set_d)), "pdgn_nndistance_grad")
    return [grad1, grad2]


class NNDistanceFunction(Function):
    @staticmethod
    def forward(ctx, seta, setb):
        ctx.save_for_backward(seta, setb)
        dist1, idx1, dist2, idx2 = NNDistance(seta, setb)
        ctx.idx1, ctx.idx2 = idx1, idx2
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2):
        seta, setb = ctx.saved_tensors
        grada, gradb = NNDistanceGrad(seta, setb, ctx.idx1, ctx.idx2, grad_dist1, grad_dist2)
        return grada, gradb


nn_distance = NNDistanceFunction.apply
