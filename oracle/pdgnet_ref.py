"""oracle/pdgnet_ref.py -- TEST INFRASTRUCTURE ONLY (torch fp32, runs on CPU).

Restatement of the floating-point part of the PDGN hot path: feature-space kNN,
edge features, the two point-deconvolution blocks, the progressive generator,
the four discriminators, the Chamfer losses and one G+D iteration.  Module
attribute names are chosen so that ``state_dict()`` keys equal the reference's
(SURVEY.md section 8-a6), which lets tests/golden/gen_golden.py load identical
weights into the real reference classes and into these.

Pinned by tests/golden/*.npz (outputs of the imported reference, see
tests/golden/gen_golden.py).  Citations are to /root/reference/models/PDGNet_v2.py
unless another file is named.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cref


# --------------------------------------------------------------------------- kNN / edges
def feature_knn(x, k):
    """:447-458 / :488-502.  x (B,F,N) -> idx (B,N,k) int64: ranks 1..k of the row-wise
    sort of  -2 x^T x + |x|^2 + |x|^2^T  (rank 0 is dropped, not "self")."""
    xt = x.transpose(1, 2)
    inner = -2 * torch.bmm(xt, x)
    sq = (xt ** 2).sum(dim=2, keepdim=True)
    dist = inner + sq + sq.transpose(1, 2)
    return dist.sort(dim=2)[1][:, :, 1:k + 1].contiguous(), dist


def edge_features(x, idx):
    """:462-477.  x (B,C,N), idx (B,N,k) -> (B,2C,N,k) = [central, neighbour - central]."""
    B, C, N = x.shape
    k = idx.shape[2]
    flat = idx.reshape(B, 1, N * k).expand(B, C, N * k)
    nbr = torch.gather(x, 2, flat).view(B, C, N, k)
    central = x.unsqueeze(3).expand(B, C, N, k)
    return torch.cat([central, nbr - central], dim=1)


# --------------------------------------------------------------------------- deconv blocks
class _ConvBnRelu(nn.Module):
    """conv2dbr :530-545 (keys conv.*, bn.*)."""

    def __init__(self, cin, cout, ksize):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, ksize, 1)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


def _interleave(inte, C, k):
    """:575-578 / :638-641.  (B,2C,N,k/2) -> (B,C,N,k): out[c, s] = conv[2c + s // (k/2), s % (k/2)]."""
    B, _, N, h = inte.shape
    return inte.transpose(2, 1).contiguous().view(B, N, C, 2 * h).permute(0, 2, 1, 3)


class EdgeDeconvRef(nn.Module):
    """upsample_edgeConv :547-588 (bilateral=False) and bilateral_upsample_edgeConv
    :590-650 (bilateral=True).  x (B,Fin,N) [, pc (B,3,N)] -> (B,Fout,2N)."""

    def __init__(self, Fin, Fout, k, bilateral, softmax=True):
        super().__init__()
        self.k, self.Fin, self.Fout = k, Fin, Fout
        self.bilateral, self.softmax = bilateral, softmax
        self.conv2 = _ConvBnRelu(2 * Fin, 2 * Fout, [1, 2 * k])
        if bilateral:
            self.conv_xyz = nn.Sequential(nn.Conv2d(6, 16, 1), nn.BatchNorm2d(16), nn.LeakyReLU())
            self.conv_fea = nn.Sequential(nn.Conv2d(2 * Fin, 16, 1), nn.BatchNorm2d(16),
                                          nn.LeakyReLU())
            self.conv_all = nn.Sequential(nn.Conv2d(16, 64, 1), nn.BatchNorm2d(64), nn.LeakyReLU(),
                                          nn.Conv2d(64, 2 * Fin, 1), nn.BatchNorm2d(2 * Fin),
                                          nn.LeakyReLU())
        self.inte_conv_hk = nn.Sequential(nn.Conv2d(2 * Fin, 4 * Fin, [1, k // 2 + 1], 1),
                                          nn.BatchNorm2d(4 * Fin), nn.LeakyReLU())

    def forward(self, x, pc=None, idx=None):
        B, Fin, N = x.shape
        if idx is None:
            idx, _ = feature_knn(x, self.k)
        e = edge_features(x, idx)
        inte = _interleave(self.inte_conv_hk(e), 2 * Fin, self.k)
        if self.bilateral:
            w = self.conv_all(self.conv_fea(e) * self.conv_xyz(edge_features(pc, idx)))
            if self.softmax:
                w = F.softmax(w, dim=-1)
            inte = inte * w
        out = self.conv2(torch.cat((e, inte), 3))            # (B, 2Fout, N, 1)
        return out.reshape(B, self.Fout, 2, N).reshape(B, self.Fout, 2 * N)


class _BilateralBlockRef(nn.Module):
    """bilateral_block_l1..l4 :672-818 folded into one class.  level 1 wraps the deconv
    in a Sequential with BN1d/LeakyReLU (keys upsample_cov.0.*, upsample_cov.1.*), levels
    2-4 use upsample_cov / bn_uc; level 4 has no g_fc branch."""

    def __init__(self, level, Fin, Fout, num_k):
        super().__init__()
        self.level = level
        k = num_k // 2
        if level == 1:
            self.upsample_cov = nn.Sequential(EdgeDeconvRef(Fin, Fout, k, False),
                                              nn.BatchNorm1d(Fout), nn.LeakyReLU())
        else:
            self.upsample_cov = EdgeDeconvRef(Fin, Fout, k, True)
            self.bn_uc = nn.BatchNorm1d(Fout)
        self.fc = nn.Sequential(nn.Linear(Fin, Fin), nn.BatchNorm1d(Fin), nn.LeakyReLU(),
                                nn.Linear(Fin, Fout), nn.BatchNorm1d(Fout), nn.LeakyReLU())
        if level < 4:
            self.g_fc = nn.Sequential(nn.Linear(Fout, 512), nn.BatchNorm1d(512), nn.LeakyReLU())

    def forward(self, x, pc=None, idx=None):
        B, _, N = x.shape
        xs = self.fc(x.max(dim=2)[0])                        # MaxPool2d((1,N)) :699-701
        if self.level == 1:
            x_ec = self.upsample_cov[2](self.upsample_cov[1](self.upsample_cov[0](x, idx=idx)))
        else:
            x_ec = F.leaky_relu(self.bn_uc(self.upsample_cov(x, pc, idx=idx)))
        x_out = torch.cat((xs.unsqueeze(2).expand(-1, -1, 2 * N), x_ec), 1)
        if self.level == 4:
            return x_out
        g = self.g_fc(xs)
        return x_out, torch.cat((g.unsqueeze(2).expand(-1, -1, 2 * N), x_ec), 1)


def _head(cin):
    """mlp1..4 :835-862."""
    return nn.Sequential(nn.Conv1d(cin, 256, 1), nn.LeakyReLU(), nn.Conv1d(256, 64, 1),
                         nn.LeakyReLU(), nn.Conv1d(64, 3, 1))


class PointGeneratorRef(nn.Module):
    """PointGenerator :820-877.  base_points=128 is the reference; 256 is the C4
    extension of SURVEY.md section 8 "Note (C4)"."""

    def __init__(self, num_k=20, base_points=128):
        super().__init__()
        self.base = base_points
        self.fc1 = nn.Sequential(nn.Linear(128, 32 * base_points), nn.BatchNorm1d(32 * base_points),
                                 nn.LeakyReLU())
        self.bilateral1 = _BilateralBlockRef(1, 32, 32, num_k)
        self.bilateral2 = _BilateralBlockRef(2, 64, 64, num_k)
        self.bilateral3 = _BilateralBlockRef(3, 128, 128, num_k)
        self.bilateral4 = _BilateralBlockRef(4, 256, 256, num_k)
        self.mlp1, self.mlp2, self.mlp3, self.mlp4 = _head(544), _head(576), _head(640), _head(512)

    def forward(self, z, idx=(None, None, None, None)):
        x = self.fc1(z).view(z.shape[0], 32, self.base)
        x1, g1 = self.bilateral1(x, idx=idx[0])
        p1 = self.mlp1(g1)
        x2, g2 = self.bilateral2(x1, p1, idx=idx[1])
        p2 = self.mlp2(g2)
        x3, g3 = self.bilateral3(x2, p2, idx=idx[2])
        p3 = self.mlp3(g3)
        p4 = self.mlp4(self.bilateral4(x3, p3, idx=idx[3]))
        return p1, p2, p3, p4


class PointDiscriminatorRef(nn.Module):
    """PointDiscriminator_1..4 :882-1023: widths (64,128,256[,512|1024]) + MLP."""

    CFG = {1: ((64, 128, 256), (128, 64)), 2: ((64, 128, 256, 512), (256, 64)),
           3: ((64, 128, 256, 512), (256, 64)), 4: ((64, 128, 256, 1024), (512, 256, 64))}

    def __init__(self, level):
        super().__init__()
        widths, hidden = self.CFG[level]
        layers, cin = [], 3
        for w in widths:
            layers += [nn.Conv1d(cin, w, 1), nn.BatchNorm1d(w), nn.LeakyReLU()]
            cin = w
        self.fc1 = nn.Sequential(*layers)
        mlp = []
        for h in hidden:
            mlp += [nn.Linear(cin, h), nn.LeakyReLU()]
            cin = h
        self.mlp = nn.Sequential(*mlp, nn.Linear(cin, 1))

    def forward(self, x):
        return self.mlp(self.fc1(x).max(dim=2)[0])


# --------------------------------------------------------------------------- losses
def chamfer_loss_sum(preds, gts):
    """utils/chamfer_loss.py:13-38.  Gram-form P (no clamp), sum of both min directions."""
    x, y = gts, preds
    zz = torch.bmm(x, y.transpose(2, 1))
    rx = (torch.bmm(x, x.transpose(2, 1))).diagonal(dim1=1, dim2=2).unsqueeze(2)
    ry = (torch.bmm(y, y.transpose(2, 1))).diagonal(dim1=1, dim2=2).unsqueeze(1)
    P = rx + ry - 2 * zz
    return P.min(1)[0].sum() + P.min(2)[0].sum()


def dist_chamfer(a, b):
    """evaluation/evaluation_metrics.py:35-45."""
    zz = torch.bmm(a, b.transpose(2, 1))
    rx = (torch.bmm(a, a.transpose(2, 1))).diagonal(dim1=1, dim2=2).unsqueeze(2)
    ry = (torch.bmm(b, b.transpose(2, 1))).diagonal(dim1=1, dim2=2).unsqueeze(1)
    P = rx + ry - 2 * zz
    return P.min(1)[0], P.min(2)[0]


def mean_covariance(points):
    """:127-134.  points (R,3,k) -> mu (R,3,1), cov (R,3,3)."""
    mu = points.mean(dim=-1, keepdim=True)
    t = points - mu
    return mu, torch.bmm(t, t.transpose(1, 2)) / points.shape[-1]


class _GroupRef(torch.autograd.Function):
    """pointops.Grouping (lib/pointops/functions/pointops.py:122-151) on the C oracle."""

    @staticmethod
    def forward(ctx, feats, idx):
        ctx.idx, ctx.n = idx, feats.shape[2]
        return torch.from_numpy(cref.grouping_forward(feats.detach().numpy(), idx.numpy()))

    @staticmethod
    def backward(ctx, g):
        return torch.from_numpy(cref.grouping_backward(g.contiguous().numpy(), ctx.idx.numpy(),
                                                       ctx.n)), None


def query_and_group_xyz(xyz, new_xyz, nsample=20):
    """Gen_QueryAndGroupXYZ pointops.py:670-703 (radius=None): knnquery + grouping."""
    idx = torch.from_numpy(cref.knnquery(nsample, xyz.detach().numpy(), new_xyz.detach().numpy())[0])
    return _GroupRef.apply(xyz.transpose(1, 2).contiguous(), idx)


def local_pair(pt1, pt2):
    """get_local_pair :136-155.  pt1 (B,3,M), pt2 (B,3,N>=M) -> (like_mu, like_cov)."""
    B, _, M = pt1.shape
    new_xyz = pt1.transpose(1, 2).contiguous()
    g1 = query_and_group_xyz(new_xyz, new_xyz).transpose(1, 2).contiguous().view(-1, 3, 20)
    g2 = query_and_group_xyz(pt2.transpose(1, 2).contiguous(), new_xyz)
    g2 = g2.transpose(1, 2).contiguous().view(-1, 3, 20)
    mu1, var1 = mean_covariance(g1)
    mu2, var2 = mean_covariance(g2)
    return (chamfer_loss_sum(mu1.view(B, -1, 3), mu2.view(B, -1, 3)) / float(M),
            chamfer_loss_sum(var1.view(B, -1, 9), var2.view(B, -1, 9)) / float(M))


# --------------------------------------------------------------------------- one G+D iteration
class TrainerRef:
    """The op sequence of PDGNet_v2.train :171-256 for one batch (no data loader, no print)."""

    def __init__(self, lr=1e-4, num_k=20, base_points=128):
        self.G = PointGeneratorRef(num_k, base_points)
        self.D = [PointDiscriminatorRef(i) for i in (1, 2, 3, 4)]
        adam = lambda m: torch.optim.Adam(m.parameters(), lr=lr, betas=(0.5, 0.999))
        self.optG, self.optD = adam(self.G), [adam(d) for d in self.D]

    def step(self, reals, z1, z2):
        """reals: 4 tensors (B,3,Nk); z1, z2 (B,128).  Returns dict of python floats."""
        B = z1.shape[0]
        ones, zeros = torch.ones(B, 1), torch.zeros(B, 1)
        mse = F.mse_loss
        fakes = self.G(z1)
        out = {}
        for i in range(4):
            self.optD[i].zero_grad()
            lossD = (mse(self.D[i](reals[i]), ones) + mse(self.D[i](fakes[i].detach()), zeros)) / 2.0
            lossD.backward()
            self.optD[i].step()
            out["d_loss%d" % (i + 1)] = lossD.item()
        self.optG.zero_grad()
        p = self.G(z2)
        sim = 0.0
        for a, b in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)):
            mu, cov = local_pair(p[a], p[b])
            sim = sim + mu + cov
        g = [mse(self.D[i](p[i]), ones) for i in range(4)]
        lossG = (1.2 * g[0] + 1.2 * g[1] + 1.2 * g[2] + g[3]) + 0.1 * sim
        lossG.backward()
        self.optG.step()
        out["g_loss"], out["similar_loss"] = lossG.item(), float(sim)
        return out
