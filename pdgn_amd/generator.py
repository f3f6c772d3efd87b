"""Progressive generator and discriminators of PDGN (models/PDGNet_v2.py:672-1023) on top of
PointDeconv.  Module / parameter names reproduce the reference's ``state_dict`` keys exactly
(tests/golden/state_dict_manifest.json), so reference checkpoints load unchanged (strip the
DataParallel ``module.`` prefix, see ``load_reference_state_dict``).

Everything between the noise vector and the emitted clouds runs POINT-MAJOR (B, N, C): a 1x1
Conv1d is a row-matrix product, a BatchNorm1d over (B,C,N) is the fused channels-last
BatchNorm+LeakyReLU kernel over B*N rows (pdgn_amd.fused.bn_act), the `cat` of a broadcast
global vector with per-point features is folded into the following layer's weights
(W [g; x] = W_g g + W_x x), and no MIOpen convolution / NCHW BatchNorm kernel is involved.
The public interfaces keep the reference's (B,C,N) layout.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import deconv as _deconv
from .deconv import PointDeconv


def _bn_act(x2d, bn, training, act="leaky_relu", pre_bias=None):
    return _deconv.bn_act(x2d, bn, training, act=act, pre_bias=pre_bias)      # looked up late: tests patch deconv.bn_act


def _small_seq(seq, x, training):
    return _deconv.small_sequential(seq, x, training)      # looked up late: tests patch deconv.small_sequential


def _point_max(x):
    from .fused import point_max
    return point_max(x)


def _linear(rows, weight, bias=None):
    return _deconv.linear_cl(rows, weight, bias)              # looked up late: tests patch deconv.linear_cl


def _linear_stats(rows, weight, training):
    """(y, BatchNorm partials of y | None): in training the GEMM's epilogue emits the statistics the BatchNorm behind it needs."""
    return _deconv.linear_cl(rows, weight, None, None, bool(training))


def _conv1x1_rows(rows, conv):
    """Conv1d(kernel 1) applied to a (rows, C_in) matrix."""
    return _linear(rows, _deconv._w2d(conv), conv.bias)


def _head_rows(head, x_rows, B, g=None, n_const=0):
    """mlp1..mlp4 (:835-862) on point-major rows.  The head's input is cat([g broadcast, x]) along
    channels (:708-709, :745-746); the first `n_const` input channels (the per-batch vector g)
    are applied once per batch and broadcast, never concatenated."""
    c0 = head[0]
    W = _deconv._w2d(c0)
    if g is not None and x_rows.is_cuda and x_rows.shape[0] >= 1024 and x_rows.shape[1] % 4 == 0 and n_const % 4 == 0:
        # on the GPU: per-sample term, activations and their derivatives in the GEMMs' epilogues (fused.HeadMLP)
        from .fused import HeadMLP
        return HeadMLP.apply(x_rows, g, W, c0.bias, _deconv._w2d(head[2]), head[2].bias, _deconv._w2d(head[4]), head[4].bias, B)
    h = _linear(x_rows, W[:, n_const:].contiguous())                       # (B*M, 256)
    if g is not None:
        M = x_rows.shape[0] // B
        h = (h.view(B, M, -1) + F.linear(g, W[:, :n_const], c0.bias).unsqueeze(1)).view(B * M, -1)
    else:
        h = h + c0.bias
    h = F.leaky_relu(h)
    h = F.leaky_relu(_conv1x1_rows(h, head[2]))
    return _conv1x1_rows(h, head[4])                                       # (B*M, 3)


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

    def forward_cl(self, xt, pct=None, idx=None, const=None, idx_ready=None):
        """xt (B,N,Fv) [, pct (B,N,3)] -> xs (B,Fout), x_ec (B,2N,Fout), g (B,512)|None.  The block's
        input is cat([const broadcast over the points (B,Fc), xt]) along channels (:708) -- passed in two
        pieces so that the broadcast half is never materialised per point."""
        B, N, _ = xt.shape
        pooled = _point_max(xt)                                 # MaxPool2d((1,N)) over the points
        if const is not None:
            pooled = torch.cat((const, pooled), 1)              # max of a broadcast channel is itself
        xs = _small_seq(self.fc, pooled, self.training)
        if self.level == 1:
            dec, bn = self.upsample_cov[0], self.upsample_cov[1]
        else:
            dec, bn = self.upsample_cov, self.bn_uc
        x_ec = dec.forward_cl(xt, pct, idx=idx, const=const, idx_ready=idx_ready)    # (B,2N,Fout)
        x_ec = _bn_act(x_ec.reshape(B * 2 * N, -1), bn, self.training).view(B, 2 * N, -1)
        g = _small_seq(self.g_fc, xs, self.training) if self.level < 4 else None
        return xs, x_ec, g

    def forward(self, x, pc=None, idx=None):
        """Reference interface: x (B,Fin,N) -> x_out (B,2Fout,2N) [, g_out (B,512+Fout,2N)]."""
        xs, x_ec, g = self.forward_cl(x.transpose(1, 2).contiguous(),
                                      pc.transpose(1, 2).contiguous() if pc is not None else None, idx=idx)
        _deconv.flush_bn_counters()
        N2 = x_ec.shape[1]
        x_ec = x_ec.transpose(1, 2)
        x_out = torch.cat((xs.unsqueeze(2).expand(-1, -1, N2), x_ec), 1)
        if g is None:
            return x_out
        return x_out, torch.cat((g.unsqueeze(2).expand(-1, -1, N2), x_ec), 1)


def _mlp_head(cin):
    """mlp1..mlp4 :835-862 (parameter container; applied through _head_rows)."""
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

    def forward(self, z, idx=(None, None, None, None), stage_hook=None):
        """`stage_hook(level, cloud)`, if given, is called as soon as the cloud of a level exists (the trainer starts
        that level's discriminator update on another stream while the deeper levels are still being generated)."""
        s = self.begin(z)
        for lvl in range(4):
            self.level(s, lvl, idx, stage_hook)
        return self.finish(s)

    # The forward in pieces (begin / level x 4 / finish): the trainer interleaves the levels of its two generator passes
    # on two streams (PDGNTrainer._step_overlapped); forward() above is the plain sequence.
    def begin(self, z):
        B = z.shape[0]
        xt = _small_seq(self.fc1, z, self.training).view(B, 32, self.base_points).transpose(1, 2).contiguous()    # (B,N0,32)
        return {"B": B, "xt": xt, "pct": None, "const": None, "clouds": [], "pending": (None, None)}

    def level(self, s, lvl, idx=(None, None, None, None), stage_hook=None):
        B = s["B"]
        blocks = (self.bilateral1, self.bilateral2, self.bilateral3, self.bilateral4)
        heads = (self.mlp1, self.mlp2, self.mlp3, self.mlp4)
        xt, pct, const, clouds = s["xt"], s["pct"], s["const"], s["clouds"]
        lvl_idx, lvl_ready = (idx[lvl], None) if idx[lvl] is not None else s["pending"]
        xs, x_ec, g = blocks[lvl].forward_cl(xt, pct, idx=lvl_idx, const=const, idx_ready=lvl_ready)
        s["pending"] = (None, None)
        if lvl < 3 and idx[lvl + 1] is None and x_ec.is_cuda:
            # the next block's kNN graph only needs this block's outputs: start it now, underneath this level's
            # MLP head, the next block's global branch and per-point GEMM
            nxt = blocks[lvl + 1].upsample_cov
            s["pending"] = _deconv.start_feature_knn(x_ec, xs, nxt.k)
        M, Fo = x_ec.shape[1], x_ec.shape[2]
        rows = x_ec.reshape(B * M, Fo)
        if lvl < 3:
            p = _head_rows(heads[lvl], rows, B, g=g, n_const=512)           # head sees cat(g, x_ec)
        else:
            p = _head_rows(heads[lvl], rows, B, g=xs, n_const=Fo)           # mlp4 sees cat(xs, x_ec) :875
        pct = p.view(B, M, 3)
        clouds.append(pct.transpose(1, 2))                                  # (B,3,M) like the reference
        if stage_hook is not None:
            _deconv.flush_bn_counters()
            stage_hook(lvl, clouds[-1])
        s["xt"], s["const"], s["pct"] = x_ec, xs, pct   # next block's input is cat(xs broadcast, x_ec) :708

    def finish(self, s):
        _deconv.flush_bn_counters()
        return tuple(s["clouds"])

    def _deconvs(self):
        blocks = (self.bilateral1, self.bilateral2, self.bilateral3, self.bilateral4)
        return [(b.upsample_cov[0] if b.level == 1 else b.upsample_cov) for b in blocks]

    def preassemble(self, deepest_without_graph=False):
        """The four blocks' re-associated GEMM operands for the current parameters, built ONCE for all the forward passes
        that follow until the parameters change (the trainer's two generator passes of an iteration).  Block l > 1 sees
        cat([xs broadcast, x_ec]): its first Fout(l-1) input channels are constant per sample (forward_cl's `const`).
        deepest_without_graph: the last block's operands are built under no_grad (they serve the no-grad pass; a pass with
        grad enabled assembles its own inside its forward -- PointDeconv.assembled)."""
        decs = self._deconvs()
        for lvl, dec in enumerate(decs):
            Fc = 0 if lvl == 0 else dec.Fin - decs[lvl - 1].Fout
            if deepest_without_graph and lvl == len(decs) - 1:
                with torch.no_grad():
                    dec.preassemble(Fc)
            else:
                dec.preassemble(Fc)

    def drop_preassembled(self):
        for dec in self._deconvs():
            dec.drop_preassembled()


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
        B, _, N = x.shape
        h = x.transpose(1, 2).reshape(B * N, 3)
        last = len(self.fc1) - 3
        for i in range(0, last, 3):                             # Conv1d(k=1) + BatchNorm1d + LeakyReLU
            # the conv bias is folded into the BatchNorm (pre_bias): the GEMM runs without a bias epilogue
            y, part = _linear_stats(h, _deconv._w2d(self.fc1[i]), self.training)
            h = _deconv.bn_act(y, self.fc1[i + 1], self.training, pre_bias=self.fc1[i].bias, partials=part)
        # last layer: BatchNorm1d + LeakyReLU + MaxPool1d(num_point) fused (the activated tensor is not written)
        w_last = _deconv._w2d(self.fc1[last])
        # (no statistics in this GEMM's epilogue: the max-pool tail takes them in the pass that finds the extremes -- fused.BNActMaxPool)
        y, part = _linear_stats(h, w_last, self.training and not _deconv.STATS_MAX)
        pooled = _deconv.bn_act_maxpool(y, self.fc1[last + 1], self.training, B, N, pre_bias=self.fc1[last].bias, partials=part,
                                        dense=_deconv.DenseInput(h, w_last))
        _deconv.flush_bn_counters()
        return _small_seq(self.mlp, pooled, self.training)            # Linear + LeakyReLU groups on B rows: one launch each


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
