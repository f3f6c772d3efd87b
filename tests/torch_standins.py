"""Test-only torch stand-ins for the HIP entry points, so the HOST logic of pdgn_amd (weight
re-association, layouts, BN bookkeeping, the G+D step, DDP) can be exercised on a CPU-only box.
They are monkeypatched in by CPU tests; the product never imports this file."""
import torch


class EdgeGatherSumTorch:
    """Same contract as pdgn_amd.deconv.EdgeGatherSum.apply, in differentiable torch ops."""

    @staticmethod
    def apply(Y, idx, specs, *biases):
        B, N, ldy = Y.shape
        idx = idx.long()
        outs, extra = [], []
        for spec, bias in zip(specs, biases):
            T, P, C, off, offc = spec[:5]
            if len(spec) > 5 and spec[5]:
                extra.append(None)                                                  # BatchNorm partials: HIP-only
            acc = Y[:, :, offc:offc + C].unsqueeze(2).expand(B, N, P, C) if offc >= 0 else 0
            for t in range(T):
                cols = Y[:, :, off + t * C: off + (t + 1) * C]                     # (B,N,C)
                nb = idx[:, :, t:t + P]                                            # (B,N,P)
                g = torch.gather(cols, 1, nb.reshape(B, N * P, 1).expand(B, N * P, C)).view(B, N, P, C)
                acc = acc + g
            if bias is not None:
                acc = acc + (bias if bias.dim() == 1 else bias.view(B, 1, 1, C))   # shared or per-sample
            outs.append(acc)
        return tuple(outs) + tuple(extra)


def feature_knn_torch(x, k):
    xt = x.transpose(1, 2)
    sq = (xt ** 2).sum(dim=2, keepdim=True)
    dist = -2 * torch.bmm(xt, x) + sq + sq.transpose(1, 2)
    return dist.sort(dim=2, stable=True)[1][:, :, 1:k + 1].to(torch.int32).contiguous()


def bn_act_torch(x2d, bn, training, act="leaky_relu", mul=None, pre_bias=None, partials=None, interleave_n=0):
    """Same contract as pdgn_amd.fused.bn_act in plain torch ops."""
    import torch.nn.functional as F
    if interleave_n:
        from pdgn_amd.fused import interleave_rows
        return interleave_rows(bn_act_torch(x2d, bn, training, act, mul, pre_bias, partials), interleave_n)
    if pre_bias is not None:
        x2d = x2d + pre_bias
    if training and bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
    y = F.batch_norm(x2d, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
    y = {"none": lambda t: t, "relu": torch.relu, "leaky_relu": F.leaky_relu}[act](y)
    return y * mul if mul is not None else y


def knnquery_oracle(nsample, xyz, new_xyz):
    from oracle import cref
    return torch.from_numpy(cref.knnquery(nsample, xyz.detach().numpy(), new_xyz.detach().numpy())[0])


def local_stats_torch(xyz, idx):
    """Same contract as pdgn_amd.losses.local_stats: gather + mean / covariance in torch ops."""
    B, N, _ = xyz.shape
    _, M, K = idx.shape
    g = torch.gather(xyz, 1, idx.long().reshape(B, M * K, 1).expand(B, M * K, 3)).view(B, M, K, 3)
    mu = g.mean(dim=2)
    t = g - mu.unsqueeze(2)
    cov = torch.einsum("bmka,bmkc->bmac", t, t) / K
    return mu, cov.reshape(B, M, 9)


def chamfer_min_torch(x, y):
    """Same contract as pdgn_amd.losses.chamfer_min (Gram-form P, minima over both axes)."""
    P = (x * x).sum(2, keepdim=True) + (y * y).sum(2).unsqueeze(1) - 2 * torch.bmm(x, y.transpose(1, 2))
    return P.min(2)[0], P.min(1)[0]


def chamfer_sum_torch(x, y, scale=1.0):
    minx, miny = chamfer_min_torch(x, y)
    return (minx.sum() + miny.sum()) * scale


def mse_const_torch(x, target, scale=1.0):
    """Same contract as pdgn_amd.losses.mse_const: scale * nn.MSELoss against a constant, in torch ops."""
    return torch.nn.functional.mse_loss(x, torch.full_like(x, float(target))) * scale


def patch_losses(monkeypatch_or_module):
    """Route pdgn_amd.losses' HIP entry points to the stand-ins above."""
    from pdgn_amd import losses
    for name, fn in (("knnquery", knnquery_oracle), ("local_stats", local_stats_torch), ("chamfer_min", chamfer_min_torch),
                     ("chamfer_sum", chamfer_sum_torch), ("mse_const", mse_const_torch)):
        if hasattr(monkeypatch_or_module, "setattr"):
            monkeypatch_or_module.setattr(losses, name, fn)
        else:
            setattr(losses, name, fn)


def linear_cl_torch(x2d, weight, bias=None, addend=None, want_stats=None, planes=None, x_max=None, x_cmax=None):
    y = torch.nn.functional.linear(x2d, weight, bias)
    y = y + addend if addend is not None else y
    return y if want_stats is None else (y, None)


def bn_softmax_slots_permute_torch(x2d, bn, training, k, act="leaky_relu", pre_bias=None):
    h = bn_act_torch(x2d, bn, training, act=act, pre_bias=pre_bias)
    return softmax_slots_permute_torch(h.view(-1, k, x2d.shape[1]))


def bilateral_weighting_torch(x2d, bn_x, u2d, bn_u, training, k, act="leaky_relu", pre_bias_x=None, pre_bias_u=None,
                              partials_u=None, partials_x=None):
    w = bn_softmax_slots_permute_torch(x2d, bn_x, training, k, act=act, pre_bias=pre_bias_x)
    return bn_act_torch(u2d, bn_u, training, act=act, mul=w.reshape(u2d.shape), pre_bias=pre_bias_u)


def softmax_slots_permute_torch(h):
    M, k, C = h.shape
    P = k // 2
    w = torch.softmax(h, dim=1)
    return w.view(M, 2, P, C).permute(0, 2, 3, 1).reshape(M, P, 2 * C)


def flush_bn_counters_noop():
    pass


def bn_act_maxpool_torch(x2d, bn, training, B, N, act="leaky_relu", pre_bias=None, partials=None, dense=None):
    return bn_act_torch(x2d, bn, training, act=act, pre_bias=pre_bias).view(B, N, -1).max(dim=1)[0]
