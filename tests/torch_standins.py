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
        outs = []
        for (T, P, C, off, offc), bias in zip(specs, biases):
            acc = Y[:, :, offc:offc + C].unsqueeze(2).expand(B, N, P, C) if offc >= 0 else 0
            for t in range(T):
                cols = Y[:, :, off + t * C: off + (t + 1) * C]                     # (B,N,C)
                nb = idx[:, :, t:t + P]                                            # (B,N,P)
                g = torch.gather(cols, 1, nb.reshape(B, N * P, 1).expand(B, N * P, C)).view(B, N, P, C)
                acc = acc + g
            if bias is not None:
                acc = acc + bias
            outs.append(acc)
        return tuple(outs)


def feature_knn_torch(x, k):
    xt = x.transpose(1, 2)
    sq = (xt ** 2).sum(dim=2, keepdim=True)
    dist = -2 * torch.bmm(xt, x) + sq + sq.transpose(1, 2)
    return dist.sort(dim=2, stable=True)[1][:, :, 1:k + 1].to(torch.int32).contiguous()


def bn_act_torch(x2d, bn, training, act="leaky_relu", mul=None):
    """Same contract as pdgn_amd.fused.bn_act in plain torch ops."""
    import torch.nn.functional as F
    if training and bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
    y = F.batch_norm(x2d, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
    y = {"none": lambda t: t, "relu": torch.relu, "leaky_relu": F.leaky_relu}[act](y)
    return y * mul if mul is not None else y
