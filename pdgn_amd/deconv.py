"""PointDeconv -- the learned point-deconvolution block of PDGN on MI355X.

Reference: ``upsample_edgeConv`` (models/PDGNet_v2.py:547-588) and
``bilateral_upsample_edgeConv`` (:590-650).  ``PointDeconv(..., bilateral=False/True)``
computes the same function with the same parameters (identical ``state_dict`` keys and
shapes) but not the same way:

  reference                                       here
  ---------                                       ----
  bmm (B,N,N) + full sort of every row            pdgn_feature_knn: MFMA Gram tile + wave top-(k+1) in LDS
  materialise e = [x_n, x_j - x_n] (B,2F,N,k)     never materialised
  Conv2d over e with [1,T] kernels                ONE per-point GEMM  Y = X^T Wcat^T  (all taps of
                                                  inte_conv_hk, conv2[:, :, :k], conv_fea at once, since
                                                  sum_t W_t (x_j - x_n) = (W_t x)_j - (W_t x)_n), followed by
                                                  pdgn_window_gather_sum over the kNN graph
  conv2 on cat(e, inte*w) (K = 2F*2k)             gather-sum half + one GEMM over inte*w only (K = 2F*k)

which removes ~3.5x of the block's FLOPs and every (B,2F,N,k)-sized edge tensor.
Activations are point-major / channels-last inside the block: (B, N, [slot,] C).
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F
from ._fn import Function

from . import _lib
from . import streams as _streams
from ._lib import check, ptr, require, stream_of
from .fused import (Planes, split_planes, skinny_linear, _zeros, bilateral_weighting, bn_act,  # noqa: F401
                    small_sequential, bn_act_maxpool, bn_softmax_slots_permute, flush_bn_counters,  # noqa: F401
                    has_zero_colsum, linear_cl, softmax_slots_permute, DenseInput, group_colsum, mark_maxima, row_maxima_buffer)
from .fused import _STATS_MAX as STATS_MAX  # noqa: F401

F32, I32 = torch.float32, torch.int32
_KNN_OVERLAP = __import__("os").environ.get("PDGN_KNN_OVERLAP", "1") == "1"
_KNN_CONST = __import__("os").environ.get("PDGN_KNN_CONST", "0") == "1"     # 1: the per-sample constant channels take part in the feature-kNN Gram

def _knn_stream(device):
    """The kNN side stream of the issuing stream (one per (device, issuing stream): concurrent generator passes must
    not share one), on a hardware queue other than the issuing stream's -- see streams.py."""
    return _streams.plan(device).knn


def _w2d(conv):
    """(C_out, C_in, 1[, 1]) kernel-1 conv weight as a (C_out, C_in) VIEW: unlike weight[:, :, 0, 0] its backward
    is a reshape, not two zero-fill + copy launches."""
    w = conv.weight
    return w.view(w.shape[0], w.shape[1])


KNN_LOG = None             # bench.py sets this to a list: every (b, f, n) Gram problem launched is appended


def feature_knn(x, k):
    """models/PDGNet_v2.py:447-458.  x (B,F,N) fp32 -> idx (B,N,k) int32 (ranks 1..k)."""
    require(x, "x", F32, 3)
    b, f, n = x.shape
    if KNN_LOG is not None:
        KNN_LOG.append((b, f, n))
    idx = torch.empty((b, n, k), dtype=I32, device=x.device)
    sq = torch.empty((b, n), dtype=F32, device=x.device)
    check(_lib.lib().pdgn_feature_knn(b, f, n, int(k), ptr(x), ptr(sq), ptr(idx), stream_of(x)),
          "pdgn_feature_knn")
    return idx


def start_feature_knn(xt, const, k, x_cf=None):
    """Launch the feature-space kNN graph of a block whose input is cat([const broadcast, xt]) (xt (B,N,Fv)
    point-major, const (B,Fc) or None).  On a GPU it is built on a second stream -- the graph is first needed by the
    gather-sum, AFTER the per-point GEMM; its selection phase is vector-ALU work next to the GEMM's matrix-core work --
    together with the transposed graph a backward pass will want.  Returns (idx, event to wait for | None)."""
    want_csr = torch.is_grad_enabled() and xt.requires_grad

    def channel_major():                                       # (B, F, N): what pdgn_feature_knn reads
        if x_cf is not None:
            return x_cf.detach().contiguous()
        x = xt.detach().transpose(1, 2)
        # The channels that are CONSTANT over a sample's points (the broadcast global vector, half of the block's input at levels
        # 2-4) do not enter the graph: they add (g_f - g_f)^2 = 0 to every pairwise distance -- in the Gram form |q|^2 + |c|^2 - 2 q.c
        # they add 2 |g|^2 - 2 |g|^2, i.e. nothing but cancellation noise, at half of the kernel's work (stage 4: 245 -> ~130 us, and the
        # graph kernel holds every CU while it runs: DESIGN.md section 10b).  The reference builds the graph from the concatenated
        # tensor (models/PDGNet_v2.py:447-458, :708): same neighbours wherever the distances are separated by more than its own
        # rounding.  PDGN_KNN_CONST=1: with them (A/B arm).
        if const is not None and _KNN_CONST:
            x = torch.cat((const.detach().unsqueeze(2).expand(-1, -1, xt.shape[1]), x), 1)
        return x.contiguous()

    with torch.no_grad():
        if not (xt.is_cuda and _KNN_OVERLAP):
            return feature_knn(channel_major(), k), None
        cur = torch.cuda.current_stream(xt.device)
        side = _knn_stream(xt.device)
        side.wait_stream(cur)
        for t in (xt, const, x_cf):                            # read on the side stream: the transpose / cat that lay the
            if t is not None:                                  # kNN input out belong to the graph's chain, not to the
                t.record_stream(side)                          # issuing stream's (2 launches per block off its queue)
        with torch.cuda.stream(side):
            x_knn = channel_major()
            idx = feature_knn(x_knn, k)
            ready = torch.cuda.Event()
            ready.record(side)                                # the consumer waits for the graph only ...
            if want_csr:
                transposed_graph(idx)                         # ... the adjoint's CSR keeps building behind it, off
                idx._pdgn_csr_ready = torch.cuda.Event()      # both critical paths; the backward waits for this event
                idx._pdgn_csr_ready.record(side)
    return idx, ready


class EdgeGatherSum(Function):
    """out_i[b,n,p,c] = bias_i[c] + Y[b,n,offc_i+c] + sum_t Y[b, idx[b,n,p+t], off_i + t*C_i + c]
    for every spec i = (T, P, C, off, offc[, want_stats]); one backward fills a single dY.  For a spec with
    want_stats the BatchNorm partial statistics of out_i (viewed as (b*n*P, C)) are returned as an extra trailing
    output (bn_act / bilateral_weighting take them as `partials`)."""

    @staticmethod
    def forward(ctx, Y, idx, specs, *biases):
        require(Y, "Y", F32, 3)
        require(idx, "idx", I32, 3)
        b, n, ldy = Y.shape
        k = idx.shape[2]
        outs, partials = [], []
        L = _lib.lib()
        for spec, bias in zip(specs, biases):
            T, P, C, off, offc = spec[:5]
            out = torch.empty((b, n, P, C), dtype=F32, device=Y.device)
            if bias is not None and bias.dim() == 2 and bias.stride(1) == 1:       # (B,C) per-sample bias, maybe a column
                bias_c, bstride = bias.detach(), bias.stride(0)                    # slice of a packed (B, sum C) tensor
            else:
                bias_c = bias.detach().contiguous() if bias is not None else None
                bstride = C if (bias is not None and bias.dim() == 2) else 0
            if len(spec) > 5 and spec[5]:
                # the consumer is a training-mode BatchNorm over (b*n*P, C): emit its partial statistics here
                L.pdgn_bn_scratch_floats.restype = ctypes.c_longlong
                scr = torch.empty(L.pdgn_bn_scratch_floats(ctypes.c_longlong(b * n * P), C), dtype=F32, device=Y.device)
                check(L.pdgn_window_gather_sum_stats(b, n, k, ldy, T, P, C, off, offc, ptr(Y), ptr(idx), ptr(bias_c),
                                                     bstride, ptr(out), ptr(scr), stream_of(Y)),
                      "pdgn_window_gather_sum_stats")
                partials.append(scr)
            else:
                check(L.pdgn_window_gather_sum(b, n, k, ldy, T, P, C, off, offc, ptr(Y), ptr(idx),
                                               ptr(bias_c), bstride, ptr(out), stream_of(Y)),
                      "pdgn_window_gather_sum")
            outs.append(out)
        specs = tuple(tuple(spec[:5]) for spec in specs)
        ctx.mark_non_differentiable(*partials)
        ctx.set_materialize_grads(False)          # the partials' "gradients" would be zero-filled tensors of their size
        ctx.specs, ctx.shape = specs, (b, n, ldy, k)
        ctx.has_bias = [0 if bias is None else bias.dim() for bias in biases]   # 0 none, 1 shared, 2 per sample
        ctx.save_for_backward(idx)
        return tuple(outs) + tuple(partials)

    @staticmethod
    def backward(ctx, *douts):
        (idx,) = ctx.saved_tensors
        b, n, ldy, k = ctx.shape
        douts = [d if d is not None else torch.zeros((b, n, P, C), dtype=F32, device=idx.device)       # an unused output
                 for d, (T, P, C, off, offc) in zip(douts, ctx.specs)]      # (trailing outputs are statistics partials)
        L = _lib.lib()
        covered = sum(T * C + (C if offc >= 0 else 0) for (T, P, C, off, offc) in ctx.specs)
        vec_ok = ldy % 4 == 0 and k <= 31 and all(C % 4 == 0 and off % 4 == 0 and (offc < 0 or offc % 4 == 0)
                                                 for (T, P, C, off, offc) in ctx.specs)
        dbias = []
        if vec_ok and covered == ldy:
            # atomic-free path: the specs tile dY completely, each element is written once
            rowptr, edges = transposed_graph(idx)
            dY = torch.empty((b, n, ldy), dtype=F32, device=idx.device)
            # a LARGE dY leaves with its ROW maxima (the kernels that write it compute them on the way): the two-part input
            # gradient of the layer below -- the per-point product, which scales dY row by row -- would scan it otherwise
            dmax = row_maxima_buffer(b * n, idx.device) if (_lib.gemm_mode() == "x2" and b * n * ldy >= (1 << 24)) else None
            first = 1
            for (T, P, C, off, offc), dout, hb in zip(ctx.specs, douts, ctx.has_bias):
                hb = 3 if hb == 1 and has_zero_colsum(dout) else hb
                dout = dout.contiguous()
                check(L.pdgn_window_gather_sum_backward_csr(b, n, k, ldy, T, P, C, off, offc, ptr(dout), ptr(rowptr),
                                                            ptr(edges), ptr(dY), ptr(dmax), first, stream_of(dout)),
                      "pdgn_window_gather_sum_backward_csr")
                first = 0
                if hb == 2 and offc >= 0:
                    # per-sample bias: sum over (n, p) of dout = sum over n of the centre columns the kernel just
                    # wrote (dY[b,j,offc+c] = sum_p dout[b,j,p,c]) -- P times less data than dout itself
                    dbias.append(group_colsum(dY.view(b * n, ldy)[:, offc:offc + C], n))
                else:
                    dbias.append(_dbias(dout, hb))
            if dmax is not None:
                mark_maxima(dY, dmax)
            return (dY, None, None) + tuple(dbias)
        dY = torch.zeros((b, n, ldy), dtype=F32, device=idx.device)
        for (T, P, C, off, offc), dout, hb in zip(ctx.specs, douts, ctx.has_bias):
            hb = 3 if hb == 1 and has_zero_colsum(dout) else hb
            dout = dout.contiguous()
            check(L.pdgn_window_gather_sum_backward(b, n, k, ldy, T, P, C, off, offc, ptr(dout), ptr(idx), ptr(dY),
                                                    stream_of(dout)), "pdgn_window_gather_sum_backward")
            dbias.append(_dbias(dout, hb))
        return (dY, None, None) + tuple(dbias)


def _dbias(dout, kind):
    if kind == 0:
        return None
    if kind == 3:                                  # feeds a training-mode BatchNorm: identically zero (fused.py)
        return _zeros((dout.shape[-1],), dout.device)
    C = dout.shape[-1]
    if dout.is_contiguous():
        return group_colsum(dout.view(-1, C))[0] if kind == 1 else group_colsum(dout.view(-1, C), dout.shape[1] * dout.shape[2])
    return dout.sum(dim=(0, 1, 2)) if kind == 1 else dout.sum(dim=(1, 2))


def transposed_graph(idx):
    """CSR of the transposed kNN graph of idx (B,N,k), memoised on the index tensor (the two
    gather-sums of a block -- features and xyz -- share one graph)."""
    cached = getattr(idx, "_pdgn_csr", None)
    if cached is not None:
        ev = getattr(idx, "_pdgn_csr_ready", None)           # built on the kNN side stream (start_feature_knn)
        if ev is not None:
            torch.cuda.current_stream(idx.device).wait_event(ev)
        return cached
    b, n, k = idx.shape
    rowptr = torch.empty((b, n + 1), dtype=I32, device=idx.device)
    edges = torch.empty((b, n * k), dtype=I32, device=idx.device)
    scratch = torch.empty((2 * b * n,), dtype=I32, device=idx.device)
    check(_lib.lib().pdgn_knn_graph_transpose(b, n, k, ptr(idx), ptr(rowptr), ptr(edges), ptr(scratch),
                                              stream_of(idx)), "pdgn_knn_graph_transpose")
    idx._pdgn_csr = (rowptr, edges)
    return idx._pdgn_csr


class SampleBias(Function):
    """Per-sample biases of a block's gather-sums under the constant-channel split, packed as (B, sum C_i):
    bb[b, o_i + c] = bias_i[c] + Yc[b, offc_i + c] + sum_t Yc[b, off_i + t*C_i + c]  (csrc/assemble.hip) -- one launch
    instead of a dozen slice / reshape / sum / add launches, one more for the adjoint."""

    @staticmethod
    def _arrays(meta):
        n = len(meta)
        mk = lambda j: (ctypes.c_int * n)(*[m[j] for m in meta])
        return n, mk(0), mk(1), mk(2), mk(3)

    @staticmethod
    def forward(ctx, Yc, meta, *biases):
        Yc = Yc.contiguous()
        B, ldy = Yc.shape
        n, T, C, off, offc = SampleBias._arrays(meta)
        bs = [b.detach().contiguous() if b is not None else None for b in biases]
        bp = (ctypes.c_void_p * n)(*[b.data_ptr() if b is not None else None for b in bs])
        bb = torch.empty((B, sum(m[1] for m in meta)), dtype=F32, device=Yc.device)
        check(_lib.lib().pdgn_sample_bias(B, ldy, n, T, C, off, offc, bp, ptr(Yc), ptr(bb), stream_of(Yc)), "pdgn_sample_bias")
        ctx.meta, ctx.shape = meta, (B, ldy)
        ctx.has = [b is not None for b in biases]
        return bb

    @staticmethod
    def backward(ctx, g):
        meta, (B, ldy) = ctx.meta, ctx.shape
        n, T, C, off, offc = SampleBias._arrays(meta)
        g = g.contiguous()
        covered = sum(m[0] * m[1] + m[1] for m in meta)
        dYc = (torch.empty if covered == ldy else torch.zeros)((B, ldy), dtype=F32, device=g.device)
        dbs = [torch.empty(m[1], dtype=F32, device=g.device) if h and ctx.needs_input_grad[2 + i] else None
               for i, (m, h) in enumerate(zip(meta, ctx.has))]
        dp = (ctypes.c_void_p * n)(*[d.data_ptr() if d is not None else None for d in dbs])
        check(_lib.lib().pdgn_sample_bias_backward(B, ldy, n, T, C, off, offc, ptr(g), ptr(dYc), dp, stream_of(g)),
              "pdgn_sample_bias_backward")
        return (dYc, None) + tuple(dbs)


class AssembleWeights(Function):
    """PointDeconv._assemble on the device: (Wi, W2, Wf | None) -> (WcatC | None, WcatV, Wb) with one gather
    kernel, and its adjoint with one more (csrc/assemble.hip) -- instead of ~15 + ~25 small torch launches."""

    @staticmethod
    def forward(ctx, Wi, W2, Wf, F_, Fo, k, T, Fc):
        Wi, W2 = Wi.contiguous(), W2.contiguous()
        Wf = Wf.contiguous() if Wf is not None else None
        P = k - T + 1
        Mw = T * 4 * F_ + 4 * F_ + k * 2 * Fo + 2 * Fo + (32 if Wf is not None else 0)
        dev = Wi.device
        WcatC = torch.empty((Mw, Fc), dtype=F32, device=dev) if Fc else None
        WcatV = torch.empty((Mw, F_ - Fc), dtype=F32, device=dev)
        Wb = torch.empty((2 * Fo, P * 4 * F_), dtype=F32, device=dev)
        check(_lib.lib().pdgn_deconv_assemble(F_, Fo, k, T, Fc, ptr(Wi), ptr(W2), ptr(Wf), ptr(WcatC), ptr(WcatV), ptr(Wb),
                                              stream_of(Wi)), "pdgn_deconv_assemble")
        ctx.cfg = (F_, Fo, k, T, Fc, Wi.shape, W2.shape, None if Wf is None else Wf.shape)
        return WcatC, WcatV, Wb

    @staticmethod
    def backward(ctx, gC, gV, gB):
        F_, Fo, k, T, Fc, si, s2, sf = ctx.cfg
        ref = gV if gV is not None else (gB if gB is not None else gC)
        gC = gC.contiguous() if gC is not None else None
        gV = gV.contiguous() if gV is not None else None
        gB = gB.contiguous() if gB is not None else None
        dWi = torch.empty(si, dtype=F32, device=ref.device)
        dW2 = torch.empty(s2, dtype=F32, device=ref.device)
        dWf = torch.empty(sf, dtype=F32, device=ref.device) if sf is not None else None
        check(_lib.lib().pdgn_deconv_assemble_backward(F_, Fo, k, T, Fc, ptr(gC), ptr(gV), ptr(gB), ptr(dWi), ptr(dW2),
                                                       ptr(dWf), stream_of(ref)), "pdgn_deconv_assemble_backward")
        return dWi, dW2, dWf, None, None, None, None, None


class _ConvBN(nn.Module):
    """Parameter container with the reference's conv2dbr keys (conv.*, bn.*) :530-545."""

    def __init__(self, cin, cout, ksize):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, ksize, 1)
        self.bn = nn.BatchNorm2d(cout)


class _XyzTaps(torch.autograd.Function):
    """conv_xyz's (16, 6) kernel over cat([p_i, p_j - p_i]) (models/PDGNet_v2.py:559-566) as the two 3-column maps of the
    re-association: rows 0-15 act on the neighbour p_j (W[:, 3:]), rows 16-31 on the centre p_i (W[:, :3] - W[:, 3:]).  Two launches
    forward and two backward (torch's cat / slice / sub autograd: eleven)."""

    @staticmethod
    def forward(ctx, Wx):
        a = Wx[:, 3:]
        return torch.cat([a, Wx[:, :3] - a], 0)

    @staticmethod
    def backward(ctx, g):
        n = g.shape[0] // 2
        g1, g2 = g[:n], g[n:]
        return torch.cat([g2, g1 - g2], 1)


class PointDeconv(nn.Module):
    """x (B,Fin,N) [, pc (B,3,N)] -> (B,Fout,2N).  ``bilateral=False``: upsample_edgeConv
    (:547-588); ``bilateral=True``: bilateral_upsample_edgeConv (:590-650)."""

    def __init__(self, Fin, Fout, k, bilateral=True, softmax=True):
        super().__init__()
        if k % 2:
            raise ValueError("k must be even (the reference's view(B,N,C,2,k//2) :576 needs it)")
        self.k, self.Fin, self.Fout = k, Fin, Fout
        self.bilateral, self.softmax = bilateral, softmax
        self.conv2 = _ConvBN(2 * Fin, 2 * Fout, [1, 2 * k])
        if bilateral:
            self.conv_xyz = nn.Sequential(nn.Conv2d(6, 16, 1), nn.BatchNorm2d(16), nn.LeakyReLU(inplace=True))
            self.conv_fea = nn.Sequential(nn.Conv2d(2 * Fin, 16, 1), nn.BatchNorm2d(16),
                                          nn.LeakyReLU(inplace=True))
            self.conv_all = nn.Sequential(nn.Conv2d(16, 64, 1), nn.BatchNorm2d(64), nn.LeakyReLU(inplace=True),
                                          nn.Conv2d(64, 2 * Fin, 1), nn.BatchNorm2d(2 * Fin),
                                          nn.LeakyReLU(inplace=True))
        self.inte_conv_hk = nn.Sequential(nn.Conv2d(2 * Fin, 4 * Fin, [1, k // 2 + 1], [1, 1]),
                                          nn.BatchNorm2d(4 * Fin), nn.LeakyReLU(inplace=True))

    # -- weight re-association (differentiable torch ops on the reference-shaped parameters)
    def _assemble(self):
        Fi, Fo, k = self.Fin, self.Fout, self.k
        T = k // 2 + 1
        Wi = self.inte_conv_hk[0].weight.squeeze(2)                   # (4F, 2F, T)
        W2 = self.conv2.conv.weight.squeeze(2)                        # (2Fo, 2F, 2k)
        W2a, W2b = W2[:, :, :k], W2[:, :, k:]
        blocks = [Wi[:, Fi:, :].permute(2, 0, 1).reshape(T * 4 * Fi, Fi),            # taps of inte_conv_hk
                  (Wi[:, :Fi, :] - Wi[:, Fi:, :]).sum(2),                             # its centre term
                  W2a[:, Fi:, :].permute(2, 0, 1).reshape(k * 2 * Fo, Fi),            # taps of conv2[..., :k]
                  (W2a[:, :Fi, :] - W2a[:, Fi:, :]).sum(2)]
        if self.bilateral:
            Wf = _w2d(self.conv_fea[0])                   # (16, 2F)
            blocks += [Wf[:, Fi:], Wf[:, :Fi] - Wf[:, Fi:]]
        Wcat = torch.cat(blocks, 0)
        P = k - T + 1
        Wb = W2b.reshape(2 * Fo, 2 * Fi, 2, P).permute(0, 3, 1, 2).reshape(2 * Fo, P * 4 * Fi)
        return Wcat, Wb, T, P

    # -- the re-associated weights of this block for the current parameter values: (WcatC | None, WcatV, Wb) from one gather
    # kernel (AssembleWeights), plus -- on the bf16 matrix cores -- the pre-split planes of the two large GEMM operands.
    # `preassemble` builds them ahead of a pass (the trainer: once per iteration, for BOTH generator passes, off the issuing
    # stream); forward_cl takes them from there while the parameters' versions still match, else builds them on the spot.
    def _weights_key(self, Fc):
        ws = [self.inte_conv_hk[0].weight, self.conv2.conv.weight] + ([self.conv_fea[0].weight, self.conv_all[3].weight] if self.bilateral else [])
        return (Fc, getattr(self, "_rows_hint", None)) + tuple((w.data_ptr(), w._version) for w in ws)

    def _assemble_now(self, Fc):
        Fi, Fo, k = self.Fin, self.Fout, self.k
        T = k // 2 + 1
        WcatC, WcatV, Wb = AssembleWeights.apply(self.inte_conv_hk[0].weight, self.conv2.conv.weight,
                                                 self.conv_fea[0].weight if self.bilateral else None, Fi, Fo, k, T, Fc)
        want_t = torch.is_grad_enabled() and WcatV.requires_grad
        with torch.no_grad():
            rows = getattr(self, "_rows_hint", None)               # B * N of the last forward: decides three bf16 / two fp16 parts
            pv = split_planes(WcatV.detach(), want_t, rows, dy_maxima_free=True)      # (its dY comes from EdgeGatherSum.backward, with maxima)
            pb = split_planes(Wb.detach(), want_t, rows, x_maxima_free=self.bilateral and self.softmax)      # (inte arrives with its maxima: bilateral_weighting's want_max)
            # the edge-level layer in front of the bilateral weighting (conv_all.3: rows * k edges x 2F x 64): forward planes only --
            # a short reduction with a wide result, the row-panel kernel's case (csrc/gemm_rp.hip), BatchNorm partials included
            # (x_maxima_free: the scan of its 64-channel input -- 4 % of the result's bytes -- is not what decides here)
            pa3 = (split_planes(_w2d(self.conv_all[3]).detach(), False, rows * self.k if rows else None, x_maxima_free=True)
                   if self.bilateral and _lib.gemm_mode() == "x2" else None)
            if pa3 is not None and pa3.parts_p != 2:
                pa3 = None
        return WcatC, WcatV, Wb, pv, pb, pa3

    def preassemble(self, Fc):
        """Build the block's GEMM operands now (under the caller's grad mode, on the current stream)."""
        self._pre = (self._weights_key(Fc), torch.is_grad_enabled(), self._assemble_now(Fc))

    def drop_preassembled(self):
        self._pre = None

    def assembled(self, Fc):
        pre = getattr(self, "_pre", None)
        if pre is not None and pre[0] == self._weights_key(Fc) and (pre[1] or not torch.is_grad_enabled()):
            return pre[2]
        return self._assemble_now(Fc)

    def forward(self, x, pc=None, idx=None):
        """Reference layout: x (B,Fin,N) [, pc (B,3,N)] -> (B,Fout,2N)."""
        out = self.forward_cl(x.transpose(1, 2).contiguous(),
                              pc.transpose(1, 2).contiguous() if pc is not None else None, idx=idx, x_cf=x)
        flush_bn_counters()
        return out.transpose(1, 2)

    def forward_cl(self, xt, pct=None, idx=None, x_cf=None, const=None, idx_ready=None):
        """Point-major layout: xt (B,N,Fv) [, pct (B,N,3)] -> (B,2N,Fout) (pre bn_uc, like the
        reference block's return value).  x_cf, if given, is the full input as (B,Fin,N).
        `const` (B,Fc): the first Fc = Fin - Fv input channels when they are constant over the points
        of a sample (the broadcast global vector xs of :704-708); their contribution to every conv is a
        per-sample vector, so they never enter the per-point GEMM (half its FLOPs at levels 2-4).
        `idx`, `idx_ready`: a kNN graph started earlier with start_feature_knn and the event that marks it complete
        (waited for right before the gather-sum)."""
        B, N, Fv = xt.shape
        Fi, Fo, k = self.Fin, self.Fout, self.k
        Fc = Fi - Fv
        training = self.training
        self._rows_hint = B * N                                        # (for the next pre-assembly: _assemble_now)
        knn_side = idx_ready
        if idx is None:
            idx, knn_side = start_feature_knn(xt, const, k, x_cf=x_cf)
        elif idx.dtype != I32:
            idx = idx.to(I32)
        idx = idx.contiguous()
        planes_v = planes_b = planes_a3 = None
        if xt.is_cuda:
            T = k // 2 + 1
            P = k - T + 1
            WcatC, WcatV, Wb, planes_v, planes_b, planes_a3 = self.assembled(Fc if const is not None else 0)
        else:                                                          # host tests: the same algebra in torch ops
            Wcat, Wb, T, P = self._assemble()
            WcatC, WcatV = (Wcat[:, :Fc], Wcat[:, Fc:].contiguous()) if const is not None else (None, Wcat)
        o_i, o_ci = 0, T * 4 * Fi
        o_a = o_ci + 4 * Fi
        o_ca = o_a + k * 2 * Fo
        o_p = o_ca + 2 * Fo
        Yc = skinny_linear(const, WcatC) if const is not None else None     # (B,Mw): per-sample contribution (csrc/skinny.hip)
        fuse_stats = training and xt.is_cuda and Fi % 4 == 0 and Fv % 4 == 0
        # the full tap layout (rows of Wcat): what the per-sample biases of the constant channels are built from
        specs_full = [(T, P, 4 * Fi, o_i, o_ci), (k, 1, 2 * Fo, o_a, o_ca)]
        biases = [self.inte_conv_hk[0].bias, self.conv2.conv.bias]
        if self.bilateral:
            specs_full.append((1, k, 16, o_p, o_p + 16))
            biases.append(self.conv_fea[0].bias)
        if Yc is not None and Yc.is_cuda:                              # bias_b = bias + centre + sum of taps of Yc, one launch
            bb = SampleBias.apply(Yc, tuple((sp[0], sp[2], sp[3], sp[4]) for sp in specs_full), *biases)
            biases = list(torch.split(bb, [sp[2] for sp in specs_full], dim=1))
        elif Yc is not None:
            biases = [bias.unsqueeze(0) + Yc[:, sp[4]:sp[4] + sp[2]] + Yc[:, sp[3]:sp[3] + sp[0] * sp[2]].reshape(B, sp[0], sp[2]).sum(1)
                      for sp, bias in zip(specs_full, biases)]       # sp = (T, P, C, off, offc)
        x2d = xt.reshape(B * N, Fv)
        specs = [(T, P, 4 * Fi, o_i, o_ci, fuse_stats)] + specs_full[1:]
        Y = linear_cl(x2d, WcatV, planes=planes_v).view(B, N, -1)      # (B,N,Mw) -- per-point GEMM
        if knn_side is not None:
            torch.cuda.current_stream(idx.device).wait_event(knn_side)
            idx.record_stream(torch.cuda.current_stream(idx.device))
        outs = EdgeGatherSum.apply(Y, idx, tuple(specs), *biases)
        inte_pre, a_pre = outs[0], outs[1]                             # (B,N,P,4F), (B,N,1,2Fo)
        part_i = outs[-1] if fuse_stats else None                      # BatchNorm partials of inte_pre
        w = None
        inte_max = inte_cmax = None                                    # inte's row / column maxima when its producer emits them (two-part conv2)
        if self.bilateral:
            Wx = _w2d(self.conv_xyz[0])                   # (16, 6)
            Yx = linear_cl(pct.reshape(B * N, 3), _XyzTaps.apply(Wx)).view(B, N, -1)
            (xyz_pre,) = EdgeGatherSum.apply(Yx, idx, ((1, k, 16, 0, 16),), self.conv_xyz[0].bias)
            xyzf = bn_act(xyz_pre.view(-1, 16), self.conv_xyz[1], training)
            h = bn_act(outs[2].view(-1, 16), self.conv_fea[1], training, mul=xyzf)   # w_fea * w_xyz :632
            # conv biases in front of a BatchNorm are not added by the GEMM (no bias epilogue / broadcast pass):
            # they cancel in the normalisation and reach only the running mean (bn_act's pre_bias)
            # in training the GEMMs' epilogues emit the BatchNorm statistics of their outputs: no statistics pass
            h, ph = linear_cl(h, _w2d(self.conv_all[0]), None, None, training)
            h = bn_act(h, self.conv_all[1], training, pre_bias=self.conv_all[0].bias, partials=ph)
            h, ph = linear_cl(h, _w2d(self.conv_all[3]), None, None, training, planes=planes_a3)
            if self.softmax:
                # conv_all.4 + LeakyReLU + softmax over the k slots + interleave w[b,n,s=P*j+p,c'] -> [b,n,p,o=2c'+j]
                # (:623-625, :634-641) AND inte = LeakyReLU(BN(inte_pre)) * w (:637, :642): one pass over both raw tensors
                if planes_b is not None and planes_b.parts_p == 2:
                    inte, inte_max, inte_cmax = bilateral_weighting(h, self.conv_all[4], inte_pre.view(-1, 4 * Fi), self.inte_conv_hk[1], training, k,
                                                                    pre_bias_x=self.conv_all[3].bias, partials_u=part_i, partials_x=ph, want_max=True)
                else:
                    inte = bilateral_weighting(h, self.conv_all[4], inte_pre.view(-1, 4 * Fi), self.inte_conv_hk[1], training, k,
                                               pre_bias_x=self.conv_all[3].bias, partials_u=part_i, partials_x=ph)
            else:
                h = bn_act(h, self.conv_all[4], training, pre_bias=self.conv_all[3].bias, partials=ph)
                w = h.view(B, N, 2, P, 2 * Fi).permute(0, 1, 3, 4, 2).reshape(B * N * P, 4 * Fi)
                inte = bn_act(inte_pre.view(-1, 4 * Fi), self.inte_conv_hk[1], training, mul=w, partials=part_i)
        else:
            # inte = LeakyReLU(BN(inte_pre))  (:637)
            inte = bn_act(inte_pre.view(-1, 4 * Fi), self.inte_conv_hk[1], training, partials=part_i)
        out_pre = linear_cl(inte.view(B * N, P * 4 * Fi), Wb, None, a_pre.view(B * N, 2 * Fo), planes=planes_b, x_max=inte_max, x_cmax=inte_cmax)    # sum in the GEMM's epilogue
        # (B,2Fout,N,1) -> view(B,Fout,2,N) -> (B,Fout,2N) (:645-647): point j*N+n of channel c is conv channel 2c+j at
        # point n; in point-major form that is (B, 2, N, Fout) -> (B, 2N, Fout), which the BatchNorm + ReLU pass stores
        # directly (interleave_n) instead of a permute copy behind it
        out = bn_act(out_pre, self.conv2.bn, training, act="relu", interleave_n=N)     # (B*2N, Fo)
        return out.view(B, 2 * N, Fo)


def upsample_edgeConv(Fin, Fout, k, num=None):
    """Constructor-compatible alias of the reference class (:552)."""
    return PointDeconv(Fin, Fout, k, bilateral=False)


def bilateral_upsample_edgeConv(Fin, Fout, k, num=None, softmax=True):
    """Constructor-compatible alias of the reference class (:595)."""
    return PointDeconv(Fin, Fout, k, bilateral=True, softmax=softmax)
