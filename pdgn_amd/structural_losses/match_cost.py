"""Approximate EMD.  Mirrors evaluation/pytorch_structural_losses/match_cost.py:6-44
(MatchCostFunction) and the pybind entry points ApproxMatch / MatchCost / MatchCostGrad
(src/structural_loss.cpp:22-78): the callee allocates outputs."""
import ctypes

import torch
from .._fn import Function

from .. import _lib
from .._lib import check, ptr, require, stream_of

F32 = torch.float32


def ApproxMatch(set_d, set_q):
    """structural_loss.cpp:22-37.  set_d (b,n,3), set_q (b,m,3) -> [match (b,m,n), temp (b,2(n+m))]."""
    require(set_d, "set_d", F32, 3)
    require(set_q, "set_q", F32, 3)
    b, n, _ = set_d.shape
    m = set_q.shape[1]
    match = torch.empty((b, m, n), dtype=F32, device=set_d.device)
    temp = torch.empty((b, (n + m) * 2), dtype=F32, device=set_d.device)
    check(_lib.lib().pdgn_approxmatch(b, n, m, ptr(set_d), ptr(set_q), ptr(match), ptr(temp),
                                      stream_of(set_d)), "pdgn_approxmatch")
    return [match, temp]


def MatchCost(set_d, set_q, match):
    """structural_loss.cpp:39-52 -> out (b)."""
    require(set_d, "set_d", F32, 3)
    require(set_q, "set_q", F32, 3)
    require(match, "match", F32, 3)
    b, n, _ = set_d.shape
    m = set_q.shape[1]
    out = torch.empty((b,), dtype=F32, device=set_d.device)
    check(_lib.lib().pdgn_matchcost(b, n, m, ptr(set_d), ptr(set_q), ptr(match), ptr(out),
                                    stream_of(set_d)), "pdgn_matchcost")
    return out


def MatchCostGrad(set_d, set_q, match):
    """structural_loss.cpp:54-69 -> [grad1 (b,n,3), grad2 (b,m,3)]."""
    require(set_d, "set_d", F32, 3)
    require(set_q, "set_q", F32, 3)
    require(match, "match", F32, 3)
    b, n, _ = set_d.shape
    m = set_q.shape[1]
    grad1 = torch.empty((b, n, 3), dtype=F32, device=set_d.device)
    grad2 = torch.empty((b, m, 3), dtype=F32, device=set_d.device)
    check(_lib.lib().pdgn_matchcost_grad(b, n, m, ptr(set_d), ptr(set_q), ptr(match), ptr(grad1),
                                         ptr(grad2), stream_of(set_d)), "pdgn_matchcost_grad")
    return [grad1, grad2]


class MatchCostFunction(Function):
    """match_cost.py:6-42: cost (b,) = sum(match * dist); backward scales MatchCostGrad."""

    @staticmethod
    def forward(ctx, seta, setb):
        ctx.save_for_backward(seta, setb)
        match, _ = ApproxMatch(seta, setb)
        ctx.match = match
        return MatchCost(seta, setb, match)

    @staticmethod
    def backward(ctx, grad_output):
        seta, setb = ctx.saved_tensors
        grada, gradb = MatchCostGrad(seta, setb, ctx.match)
        g = grad_output.unsqueeze(1).unsqueeze(2)
        return grada * g, gradb * g


def emd_cost(seta, setb):
    """Forward-only fused EMD cost (no (b,m,n) match matrix in HBM): what the eval path needs
    (evaluation/evaluation_metrics.py:26-31 runs under no_grad and drops `match`)."""
    require(seta, "seta", F32, 3)
    require(setb, "setb", F32, 3)
    b, n, _ = seta.shape
    m = setb.shape[1]
    L = _lib.lib()
    L.pdgn_emd_cost_temp_floats.restype = ctypes.c_longlong
    temp = torch.empty(L.pdgn_emd_cost_temp_floats(ctypes.c_longlong(b), n, m), dtype=F32, device=seta.device)
    out = torch.empty((b,), dtype=F32, device=seta.device)
    check(_lib.lib().pdgn_emd_cost(b, n, m, ptr(seta), ptr(setb), ptr(temp), ptr(out), stream_of(seta)),
          "pdgn_emd_cost")
    return out


def match_cost(seta, setb):
    """match_cost.py:44.  Uses the fused forward when no gradient is required."""
    if torch.is_grad_enabled() and (seta.requires_grad or setb.requires_grad):
        return MatchCostFunction.apply(seta, setb)
    return emd_cost(seta, setb)
