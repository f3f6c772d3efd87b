"""Progressive generator and discriminators of PDGN (models/PDGNet_v2.py:672-1023) on top of
PointDeconv.  Module / parameter names reproduce the reference's ``state_dict`` keys exactly
(tests/golden/state_dict_manifest.json), so reference checkpoints load unchanged (strip the
DataParallel ``module.`` prefix, see ``load_reference_state_dict``).

The deconvolution blocks run on the hand-written HIP path (pdgn_amd.deconv); the small dense
layers around them (Linear / 1x1 Conv1d / BatchNorm1d of the global branch, the MLP heads and
the PointNet-style discriminators) are plain library GEMMs through PyTorch-ROCm.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .deconv import PointDeconv


class BilateralBlock(nn.Module):
    """bilateral_block_l1 (:672-711), _l2 (:713-748), _l3 (:750-789), _l4 (:791-818).
    level 1: plain deconv inside a Sequential with BN1d/LeakyReLU; levels 2-4: bilateral deconv +
    bn_uc; level 4 has no global `g_fc` branch."""

    def __init__(self, level, Fin, Fout, num_k=20, softmax=True):
        super().__init__()
        self.level = level
        if level == 1:
            self.upsample_cov = nn.Sequential(PointDeconv(Fin, Fout, num_k // 2, bilateral=False),
                                              nn.BatchNorm1d(Fout), nn.LeakyReLU(inplace=True))
        else:
            self.upsample_cov = PointDeconv(Fin, Fout, num_k // 2, bilateral=True, softmax=softmax)
            self.bn_uc = nn.BatchNorm1d(Fout)
        self.fc = nn.Sequential(nn.Linear(Fin, Fin), nn.BatchNorm1d(Fin), nn.LeakyReLU(inplace=True),
                                nn.Linear(Fin, Fout), nn.BatchNorm1d(Fout), nn.LeakyReLU(inplace=True))
        if level < 4:
            self.g_fc = nn.Sequential(nn.Linear(Fout, 512), nn.BatchNorm1d(512), nn.LeakyReLU(inplace=True))

    def forward(self, x, pc=None, idx=None):
        N2 = 2 * x.shape[2]
        xs = self.fc(torch.amax(x, dim=2))                      # MaxPool2d((1,N)) over the points
        if self.level == 1:
            x_ec = self.upsample_cov[2](self.upsample_cov[1](self.upsample_cov[0](x, idx=idx)))
        else:
            x_ec = F.leaky_relu(self.bn_uc(self.upsample_cov(x, pc, idx=idx)))
        x_out = torch.cat((xs.unsqueeze(2).expand(-1, -1, N2), x_ec), 1)
        if self.level == 4:
            return x_out
        g = self.g_fc(xs)
        return x_out, torch.cat((g.unsqueeze(2).expand(-1, -1, N2), x_ec), 1)


def _mlp_head(cin):
    """mlp1..mlp4 :835-862."""
    return nn.Sequential(nn.Conv1d(cin, 256, 1), nn.LeakyReLU(inplace=True), nn.Conv1d(256, 64, 1),
                         nn.LeakyReLU(inplace=True), nn.Conv1d(64, 3, 1, bias=True))


class PointGenerator(nn.Module):
    """PointGenerator :820-877: z (B,128) -> four clouds (B,3,2b), (B,3,4b), (B,3,8b), (B,3,16b)
    with b = base_points.  base_points=128 is the reference (256..2048 points); 256 gives the
    "4-stage 256->4096" configuration of BASELINE.json (Note C4).
    `num_point` is accepted and ignored like in the reference (:823)."""

    def __init__(self, num_point=2048, num_k=20, softmax=True, base_points=128):
        super().__init__()
        self.num_point, self.num_k, self.base_points = num_point, num_k, base_points
        self.fc1 = nn.Sequential(nn.Linear(128, 32 * base_points), nn.BatchNorm1d(32 * base_points),
                                 nn.LeakyReLU(inplace=True))
        self.bilateral1 = BilateralBlock(1, 32, 32, num_k)
        self.bilateral2 = BilateralBlock(2, 64, 64, num_k, softmax)
        self.bilateral3 = BilateralBlock(3, 128, 128, num_k, softmax)
        self.bilateral4 = BilateralBlock(4, 256, 256, num_k, softmax)
        self.mlp1, self.mlp2 = _mlp_head(512 + 32), _mlp_head(512 + 64)
        self.mlp3, self.mlp4 = _mlp_head(512 + 128), _mlp_head(512)

    def forward(self, z, idx=(None, None, None, None)):
        x = self.fc1(z).view(z.shape[0], 32, self.base_points)
        x1, g1 = self.bilateral1(x, idx=idx[0])
        x1s = self.mlp1(g1)
        x2, g2 = self.bilateral2(x1, x1s, idx=idx[1])
        x2s = self.mlp2(g2)
        x3, g3 = self.bilateral3(x2, x2s, idx=idx[2])
        x3s = self.mlp3(g3)
        x4s = self.mlp4(self.bilateral4(x3, x3s, idx=idx[3]))
        return x1s, x2s, x3s, x4s


class PointDiscriminator(nn.Module):
    """PointDiscriminator_1..4 :882-1023.  (B,3,N) -> (B,1)."""

    WIDTHS = {1: (64, 128, 256), 2: (64, 128, 256, 512), 3: (64, 128, 256, 512), 4: (64, 128, 256, 1024)}
    HIDDEN = {1: (128, 64), 2: (256, 64), 3: (256, 64), 4: (512, 256, 64)}

    def __init__(self, level, num_point=None):
        super().__init__()
        self.level = level
        self.num_point = num_point if num_point is not None else 128 << level
        layers, cin = [], 3
        for w in self.WIDTHS[level]:
            layers += [nn.Conv1d(cin, w, 1), nn.BatchNorm1d(w), nn.LeakyReLU(inplace=True)]
            cin = w
        self.fc1 = nn.Sequential(*layers)
        mlp = []
        for h in self.HIDDEN[level]:
            mlp += [nn.Linear(cin, h), nn.LeakyReLU(inplace=True)]
            cin = h
        mlp.append(nn.Linear(cin, 1))
        self.mlp = nn.Sequential(*mlp)

    def forward(self, x):
        return self.mlp(torch.amax(self.fc1(x), dim=2))       # MaxPool1d(num_point) over all points


def PointDiscriminator_1(num_point=256):
    return PointDiscriminator(1, num_point)


def PointDiscriminator_2(num_point=512):
    return PointDiscriminator(2, num_point)


def PointDiscriminator_3(num_point=1024):
    return PointDiscriminator(3, num_point)


def PointDiscriminator_4(num_point=2048):
    return PointDiscriminator(4, num_point)


def load_reference_state_dict(module, state_dict):
    """Load a reference checkpoint entry (``G_model`` / ``D_model1..4`` of the files written by
    models/PDGNet_v2.py:384-408); keys carry nn.DataParallel's ``module.`` prefix (:101-105)."""
    clean = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
    return module.load_state_dict(clean, strict=True)
