"""Autograd wrappers of the fused channels-last kernels of libpdgn_hip.so (bnact.hip)."""
import ctypes
import os
import weakref

import torch
from ._fn import Function

from . import _lib
from ._lib import check, ptr, require, stream_of

F32 = torch.float32
ACT = {"none": 0, "relu": 1, "leaky_relu": 2}


def _scratch_floats(L, rows, C):
    L.pdgn_bn_scratch_floats.restype = ctypes.c_longlong
    n = L.pdgn_bn_scratch_floats(ctypes.c_longlong(rows), C)
    if n < 0:
        raise _lib.PdgnHipError("pdgn_bn_scratch_floats: argument outside the supported range")
    return n


# The gradient of a training-mode BatchNorm with respect to its input has zero column sums:
#   dx = g*rstd*(dz - mean(dz) - xhat*mean(dz*xhat))  =>  sum_rows dx = -g*rstd*mean(dz*xhat)*sum_rows(xhat) = 0,
# so the bias of the conv / linear layer that FEEDS the BatchNorm (all conv2dbr-style pairs of the reference)
# has an identically zero gradient; what a row-sum of dx would return is fp32 rounding residue.  The BatchNorm
# backward marks its dx, and the producer's backward (LinearCL, EdgeGatherSum) skips that full pass over dy.
# Keyed by data pointer, validated through a weak reference to the marked tensor (same object or a view of it).
_ZERO_COLSUM = {}


def mark_zero_colsum(dx):
    _ZERO_COLSUM[dx.data_ptr()] = weakref.ref(dx)


def has_zero_colsum(dy):
    ref = _ZERO_COLSUM.pop(dy.data_ptr(), None)
    src = ref() if ref is not None else None
    return src is not None and (src is dy or dy._base is src) and dy.numel() == src.numel() \
        and dy.shape[-1] == src.shape[-1]


def clear_zero_colsum():
    _ZERO_COLSUM.clear()
    _INPUT_GRADS.clear()
    _GRAD_MAXIMA.clear()


# A third side channel of the same kind: a backward that WRITES a large gradient (EdgeGatherSum's dY, 1.8 GB at stage 4) leaves its
# ROW maxima (an int32 (rows,) tensor of bit patterns, csrc/wgs.hip) for the two-part contraction of the layer that receives it
# (LinearCL.backward: the input gradient of the per-point product scales dY row by row) -- it would scan the whole tensor otherwise.
# Keyed and validated like _ZERO_COLSUM; taken once.
_GRAD_MAXIMA = {}


def mark_maxima(g, slot):
    _GRAD_MAXIMA[g.data_ptr()] = (weakref.ref(g), slot)


def take_maxima(dy):
    ent = _GRAD_MAXIMA.pop(dy.data_ptr(), None)
    if ent is None:
        return None
    src = ent[0]()
    # the same elements: the marked tensor itself, or a view of the same storage that starts where it starts, has as many
    # elements and is, like it, contiguous (a reshape)
    same = src is not None and (src is dy or (dy.numel() == src.numel() and dy.is_contiguous() and src.is_contiguous()
                                              and (dy._base is src or (dy._base is not None and dy._base is src._base))))
    return ent[1] if same else None


# A second side channel of the same kind: BNActMaxPool's backward, when the dense layer in front of it is frozen, computes
# that layer's INPUT gradient itself (pdgn_dense_bn_maxpool_input_grad: the dense (rows, C) dx is never formed) and hands
# LinearCL's backward a zero-stride placeholder of dx's shape; the input gradient travels here, keyed like _ZERO_COLSUM.
_INPUT_GRADS = {}
_CLOSED_TAIL = os.environ.get("PDGN_CLOSED_TAIL", "1") == "1"    # A/B switch
_STATS_MAX = os.environ.get("PDGN_STATS_MAX", "1") == "1"        # A/B switch: statistics + extremes of the max-pool tail in one pass
_CLOSED_DW = os.environ.get("PDGN_CLOSED_DW", "1") == "1"        # A/B switch: also with a trainable layer (its weight gradient)


class DenseInput:
    """The (input rows, weight) of the LinearCL whose output a bn_act_maxpool consumes -- not an autograd edge."""

    def __init__(self, h, w):
        self.h, self.w = h, w


_NAN_SLOTS = {}                                                  # per device: (256 NaNs, next slot)


def _placeholder_with_input_grad(rows, C, dh, dw, device):
    """A zero-stride (rows, C) view of ONE NaN: whoever reads it as a gradient -- anything but the LinearCL it is meant for --
    produces NaNs, not plausible numbers.  The slots rotate so that placeholders pending at the same time (the
    four discriminators inside one generator backward) have distinct keys."""
    pool = _NAN_SLOTS.get(device)
    if pool is None:
        pool = _NAN_SLOTS[device] = [torch.full((256,), float("nan"), dtype=F32, device=device), 0]
    i = pool[1]
    pool[1] = (i + 1) % 256
    tok = pool[0][i:i + 1].expand(rows, C)
    _INPUT_GRADS[tok.data_ptr()] = (weakref.ref(tok), (dh, dw))
    return tok


def is_placeholder(dy):
    return dy.dim() == 2 and dy.stride(0) == 0 and dy.stride(1) == 0 and dy.numel() > 1


def take_input_grad(dy):
    hit = _INPUT_GRADS.pop(dy.data_ptr(), None) if dy.stride(0) == 0 else None
    if hit is None:
        return None
    src = hit[0]()
    return hit[1] if src is not None and (src is dy or dy._base is src or dy._base is src._base) and dy.shape == src.shape else None


# gemm_tn accumulates split partial sums with atomics, so every weight gradient starts from zeros (and the
# analytically-zero bias gradients are zeros): ~90 tiny fill launches per iteration.  A caller that brackets its
# backward passes (FlatGrads.begin) gets them as slices of ONE zero-filled arena, sized from the previous pass.
_ARENA = {"buf": None, "off": 0, "need": 0, "owner": None}


def reset_zero_arena(device, owner):
    """Start a backward pass of `owner` (any object; its attribute `zero_arena_floats` remembers how many zero
    floats its previous pass asked for)."""
    a = _ARENA
    if a["owner"] is not None:
        a["owner"].zero_arena_floats = max(getattr(a["owner"], "zero_arena_floats", 0), a["need"])
    size = getattr(owner, "zero_arena_floats", 0)
    a["buf"] = torch.zeros(size, dtype=F32, device=device) if size else None
    a["off"] = a["need"] = 0
    a["owner"] = owner


def release_zero_arena():
    """Drop the current arena (end of a step / of a graph capture): a backward outside FlatGrads.begin() then gets
    ordinary torch.zeros tensors, never slices of memory another owner (or a hipGraph's private pool) holds."""
    a = _ARENA
    if a["owner"] is not None:
        a["owner"].zero_arena_floats = max(getattr(a["owner"], "zero_arena_floats", 0), a["need"])
    a["buf"] = None
    a["owner"] = None
    a["off"] = a["need"] = 0


def _zeros(shape, device):
    n = 1
    for d in shape:
        n *= d
    step = (n + 63) // 64 * 64
    a = _ARENA
    a["need"] += step
    buf = a["buf"]
    if buf is None or buf.device != device or a["off"] + step > buf.numel():
        return torch.zeros(shape, dtype=F32, device=device)
    v = buf[a["off"]:a["off"] + n].view(shape)
    a["off"] += step
    return v


_GROUP_COLSUM = os.environ.get("PDGN_GROUP_COLSUM", "1") == "1"  # A/B switch: 0 = torch's sum(dim=...) for the bias gradients


def group_colsum(x2d, group_rows=None):
    """Column sums of x2d (rows, C) per group of `group_rows` consecutive rows (None: all rows) -> (rows / group_rows, C), on
    pdgn_group_colsum (one launch into the backward pass's zero arena); torch's reduction for shapes it does not take."""
    rows, C = x2d.shape
    gr = rows if group_rows is None else int(group_rows)
    if not (_GROUP_COLSUM and x2d.is_cuda and x2d.dtype == F32 and x2d.stride(1) == 1 and x2d.stride(0) % 4 == 0
            and x2d.data_ptr() % 16 == 0 and 16 <= C <= 1024 and (C & (C - 1)) == 0 and gr > 0 and rows % gr == 0
            and rows // gr <= 65535):
        return x2d.sum(dim=0, keepdim=True) if gr == rows else x2d.reshape(rows // gr, gr, C).sum(dim=1)
    out = _zeros((rows // gr, C), x2d.device)
    check(_lib.lib().pdgn_group_colsum(ctypes.c_longlong(rows // gr), ctypes.c_longlong(gr), C, ptr(x2d), ctypes.c_longlong(x2d.stride(0)),
                                       ptr(out), stream_of(x2d)), "pdgn_group_colsum")
    return out


def _bn_stats(L, x, rows, C, g, b, pre_bias, running_mean, running_var, training, momentum, eps, partials=None):
    """[scale|shift|mean|invstd] of a BatchNorm over x (rows, C) -- batch statistics (+ running-stat update) in
    training, running statistics in eval.  pre_bias: see include/pdgn_hip.h (the producer's bias, left out of x)."""
    stats = torch.empty(4 * C, dtype=F32, device=x.device)
    pb = pre_bias.detach().contiguous() if pre_bias is not None else None
    block = None
    if isinstance(partials, tuple):                             # (tensor (nparts, 3C), rows per block): a GEMM / thin-layer epilogue's
        partials, block = partials                              # block-shifted rows, with the block size linear_cl attached
    if training and block is not None:
        nparts = partials.shape[0]
        L.pdgn_bn_blocks_scratch_doubles.restype = ctypes.c_longlong
        nd = L.pdgn_bn_blocks_scratch_doubles(C, ctypes.c_longlong(nparts))
        scr = torch.empty(nd, dtype=torch.float64, device=x.device) if nd > 0 else None
        check(L.pdgn_bn_stats_from_gemm_partials(ctypes.c_longlong(rows), C, ctypes.c_longlong(nparts), block,
                                                 ctypes.c_float(eps), ctypes.c_float(momentum), ptr(g), ptr(b), ptr(pb),
                                                 ptr(running_mean), ptr(running_var), ptr(partials), ptr(stats), ptr(scr),
                                                 stream_of(x)),
              "pdgn_bn_stats_from_gemm_partials")
    elif training and partials is not None:                     # first stage done by x's producer (its epilogue)
        check(L.pdgn_bn_stats_from_partials(ctypes.c_longlong(rows), C, ctypes.c_float(eps), ctypes.c_float(momentum),
                                            ptr(g), ptr(b), ptr(pb), ptr(running_mean), ptr(running_var), ptr(partials),
                                            ptr(stats), stream_of(x)), "pdgn_bn_stats_from_partials")
    elif training:
        scratch = torch.empty(_scratch_floats(L, rows, C), dtype=F32, device=x.device)
        check(L.pdgn_bn_stats(ctypes.c_longlong(rows), C, ctypes.c_float(eps), ctypes.c_float(momentum), ptr(x),
                              ptr(g), ptr(b), ptr(pb), ptr(running_mean), ptr(running_var), ptr(scratch), ptr(stats),
                              stream_of(x)), "pdgn_bn_stats")
    else:
        check(L.pdgn_bn_eval_stats(C, ctypes.c_float(eps), ptr(g), ptr(b), ptr(pb), ptr(running_mean),
                                   ptr(running_var), ptr(stats), stream_of(x)), "pdgn_bn_eval_stats")
    return stats


_NO_FOLD = os.environ.get("PDGN_FOLD_BIAS", "1") == "0"       # A/B switch for benchmarking only


def _fold_pre_bias(x2d, pre_bias, training):
    """The fused kernels fold a producer bias only where its gradient is known without a pass over dx: training mode
    (identically zero) or no gradient needed.  Otherwise it is added explicitly and autograd does the rest."""
    if pre_bias is not None and (_NO_FOLD or (not training and torch.is_grad_enabled() and pre_bias.requires_grad)):
        return x2d + pre_bias, None
    return x2d, pre_bias


def _pre_bias_grad(ctx_has, C, device):
    return _zeros((C,), device) if ctx_has else None


class BNActCL(Function):
    """y = act(BatchNorm(x)) [* mul] for a channels-last (rows, C) matrix, with nn.BatchNorm
    semantics (batch statistics + running-stat update in training, running statistics in eval).
    One Function = the reference's BatchNorm2d + LeakyReLU/ReLU pair (and, with ``mul``, the
    bilateral product ``inte_x * w`` of models/PDGNet_v2.py:642)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, act, mul, pre_bias=None,
                partials=None, interleave_n=0):
        rows, C = x.shape
        x = x.contiguous()
        L = _lib.lib()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        stats = _bn_stats(L, x, rows, C, g, b, pre_bias, running_mean, running_var, training, momentum, eps, partials)
        ctx.has_pre_bias = pre_bias is not None
        mul_c = mul.contiguous() if mul is not None else None
        # interleave_n = N: y is (rows * 2, C / 2), x row b*N + n / channel 2c + j at y row b*2N + j*N + n / channel c
        y = torch.empty((rows * 2, C // 2), dtype=F32, device=x.device) if interleave_n else torch.empty_like(x)
        check(L.pdgn_bn_act_forward(ctypes.c_longlong(rows), C, act, ptr(x), ptr(stats), ptr(mul_c), ptr(y), int(interleave_n),
                                    stream_of(x)), "pdgn_bn_act_forward")
        ctx.save_for_backward(x, stats, mul_c)
        ctx.cfg = (rows, C, act, bool(training), mul is not None and mul.requires_grad, int(interleave_n))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, mul = ctx.saved_tensors
        rows, C, act, training, need_dmul, inter = ctx.cfg
        dy = dy.contiguous()
        L = _lib.lib()
        scratch = torch.empty(_scratch_floats(L, rows, C), dtype=F32, device=x.device)
        bs = torch.empty(2 * C, dtype=F32, device=x.device)
        dx = torch.empty_like(x)
        dmul = torch.empty_like(x) if need_dmul else None
        check(L.pdgn_bn_act_backward(ctypes.c_longlong(rows), C, act, int(training), ptr(x), ptr(dy), ptr(mul),
                                     ptr(stats), ptr(scratch), ptr(bs), ptr(dx), ptr(dmul), inter, stream_of(x)),
              "pdgn_bn_act_backward")
        if training:
            mark_zero_colsum(dx)
        return (dx, bs[C:], bs[:C], None, None, None, None, None, None, dmul, _pre_bias_grad(ctx.has_pre_bias, C, x.device),
                None, None)


# nn.BatchNorm's num_batches_tracked bookkeeping: one tiny int64 add per layer per forward would be
# ~100 launches per iteration; increments are collected here and applied by flush_bn_counters() with
# one multi-tensor add (the network-level forward()s and the trainer call it).
_PENDING_COUNTS = {}
_HOLD_COUNTS = [False]


def hold_bn_counters(on):
    """While held, flush_bn_counters() collects only: a caller that brackets a whole training iteration (PDGNTrainer's
    overlapped step) applies the iteration's increments with ONE launch at its end instead of one per network-level forward
    (~30 per iteration, on whatever stream the forward ran).  Nothing on the device reads the counters.  Returns the old state."""
    old, _HOLD_COUNTS[0] = _HOLD_COUNTS[0], bool(on)
    return old


def flush_bn_counters():
    if _HOLD_COUNTS[0]:
        return
    if _PENDING_COUNTS:
        tensors = list(_PENDING_COUNTS.keys())
        torch._foreach_add_(tensors, [int(v) for v in _PENDING_COUNTS.values()])
        _PENDING_COUNTS.clear()


def bn_act(x2d, bn, training, act="leaky_relu", mul=None, pre_bias=None, partials=None, interleave_n=0):
    """Apply an nn.BatchNorm{1,2}d module's parameters/buffers to a channels-last (rows, C) view,
    followed by `act` (and an optional elementwise product).  `pre_bias`: the bias of the layer that produced
    x2d, when the caller did not add it (it cancels inside the BatchNorm; only the running mean sees it)."""
    if training and bn.track_running_stats:
        _PENDING_COUNTS[bn.num_batches_tracked] = _PENDING_COUNTS.get(bn.num_batches_tracked, 0) + 1
    x2d, pre_bias = _fold_pre_bias(x2d, pre_bias, training)
    if x2d.shape[1] % 4:                      # odd channel counts: library path
        if pre_bias is not None:
            x2d = x2d + pre_bias
        y = torch.nn.functional.batch_norm(x2d, bn.running_mean, bn.running_var, bn.weight, bn.bias, training,
                                           bn.momentum, bn.eps)
        y = {"none": lambda t: t, "relu": torch.relu, "leaky_relu": torch.nn.functional.leaky_relu}[act](y)
        y = y * mul if mul is not None else y
        return interleave_rows(y, interleave_n) if interleave_n else y
    if interleave_n and (mul is not None or not x2d.is_cuda):
        return interleave_rows(bn_act(x2d, bn, training, act, mul, pre_bias, partials), interleave_n)
    return BNActCL.apply(x2d, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps,
                         ACT[act], mul, pre_bias, partials, interleave_n)


def interleave_rows(y2d, n):
    """(B*n, 2F) -> (B*2n, F): row b*n + p, channel 2c + j  ->  row b*2n + j*n + p, channel c (models/PDGNet_v2.py:645-647 in
    point-major form; the torch form of what BNActCL's interleave_n store does)."""
    rows, c2 = y2d.shape
    return y2d.view(rows // n, n, c2 // 2, 2).permute(0, 3, 1, 2).reshape(rows * 2, c2 // 2)


# ---------------------------------------------------------------------------------------------------------------
# Dense layers on point-major rows.  Every contraction with >= _OWN_MIN_ROWS rows runs on the hand-written matrix-core
# kernels of libpdgn_hip.so: forward y = x W^T and input gradient dx = dy W on pdgn_gemm_nt / pdgn_gemm_nn (csrc/gemm_x3.hip:
# fp32 operands, products on the bf16 matrix cores; PDGN_GEMM=fp32: csrc/gemm_nt.hip, the fp32 matrix instructions), the
# weight gradient dW = dy^T x on pdgn_gemm_tn_big (the same kernels) or pdgn_gemm_tn (csrc/gemm_tn.hip, small outputs).
# No BLAS library, no run-time back-end selection.  Below that row count (the 35-row per-sample layers) torch's
# default matmul is used as it is.
_OWN_MIN_ROWS = 1024
_TN_BIG = os.environ.get("PDGN_TN_BIG", "1") == "1"            # A/B switch: weight gradients of >= 128 x 64 outputs on pdgn_gemm_tn_big


def _tn_big_max():
    # x3: also the two largest outputs (measured 0.88x / 0.90x pdgn_gemm_tn's time); fp32 instructions: up to 1 M elements
    return (1 << 22) if _lib.matrix_core_mode() else (1 << 20)


def _pad_cols(t, mult=4):
    """(rows, c) -> contiguous (rows, ceil(c / mult) * mult), zero-padded; t itself when nothing has to change."""
    c = t.shape[1]
    if c % mult == 0 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
        return t
    if c % mult == 0:
        return t.contiguous()
    return torch.nn.functional.pad(t, (0, mult - c % mult)).contiguous()


GEMM_LOG = None            # tools/gemm_shapes.py sets this to a list: every (kind, m, n, k) launched is appended


_PLANES = os.environ.get("PDGN_PLANES", "1") == "1"            # A/B switch: 0 = every operand split in the kernel's loaders


class Planes:
    """A weight's parts as planes, written ONCE per weight instead of by every workgroup's loader of the contraction kernel:
    three bf16 parts (x = h + m + l, csrc/split.hip) as [3][rows][ld], or -- for a contraction that runs on two parts
    (pdgn_gemm_two_part) -- two fp16 parts [2][rows][ld], every row scaled by its own power of two, with the rows' maxima behind
    them (int16 storage either way);
    `t` = the same for the transpose (the input-gradient product dX = dY W is the NT product against W^T) or None.
    parts_p / parts_t: 3 or 2."""
    __slots__ = ("p", "t", "shape", "parts_p", "parts_t")

    def __init__(self, p, t, shape, parts_p=3, parts_t=3):
        self.p, self.t, self.shape, self.parts_p, self.parts_t = p, t, shape, parts_p, parts_t


def two_part(m, n, k, scan_bytes):
    """Whether the contraction (m, n, k) runs on two scaled fp16 parts when scan_bytes of its operands still have to be scanned
    for their maxima (mode "x2" and a launch whose time is its matrix-core work: csrc/gemm_x3.hip x2_pays)."""
    return _lib.gemm_mode() == "x2" and bool(_lib.lib().pdgn_gemm_two_part(ctypes.c_longlong(m), n, k, ctypes.c_longlong(scan_bytes)))


def two_part_planes(m, n, k, scan_bytes):
    """Whether the product (m, n, k) against PRE-SPLIT planes should use two-part ones when scan_bytes of the activations still have
    to be scanned (csrc/gemm_x3.hip pdgn_gemm_two_part_planes: from ~2 GFLOP on, the 256 x 128 eight-wave tile)."""
    return _lib.gemm_mode() == "x2" and bool(_lib.lib().pdgn_gemm_two_part_planes(ctypes.c_longlong(m), n, k, ctypes.c_longlong(scan_bytes)))


def _slots(t):
    """Before a contraction call: in the default mode the library's scale ring must live on the tensor's device (_lib.ensure_scale_slots)."""
    if _lib.gemm_mode() == "x2":
        _lib.ensure_scale_slots(t.device)


def split_planes(w, want_t, rows=None, dy_maxima_free=False, x_maxima_free=False):
    """The pre-split planes of a (n, k) fp32 weight (Planes); None where the pre-split path does not apply (fp32-instruction mode,
    sizes that would need padding).  rows: the row count of the activations it will multiply, when known -- it decides between
    three bf16 and two fp16 parts per product (forward: (rows, n, k); input gradient through the transpose: (rows, k, n));
    dy_maxima_free / x_maxima_free: the layer's output gradient / its input arrives with its maxima (mark_maxima, linear_cl's
    x_max: nothing to scan)."""
    _slots(w)
    mode = _lib.gemm_mode()
    if (not (_PLANES and w.is_cuda and w.dim() == 2 and mode != "fp32") or w.shape[0] % 4 or w.shape[1] % 4 or w.stride(1) != 1
            or w.stride(0) % 4 or w.data_ptr() % 16):
        return None
    n, k = w.shape
    ldp, ldt = (k + 7) // 8 * 8, (n + 7) // 8 * 8
    parts_p = 2 if rows and two_part_planes(rows, n, k, 0 if x_maxima_free else rows * k * 4) else 3    # (the weight brings its exponent)
    parts_t = 2 if rows and want_t and two_part_planes(rows, k, n, 0 if dy_maxima_free else rows * n * 4) else 3

    def planes(parts, rows_, ld):                                  # [parts][rows][ld] (+ 4 B per row: two-part planes keep their rows' maxima there)
        buf = torch.empty(parts * rows_ * ld + 2 * rows_ + 8, dtype=torch.int16, device=w.device)
        return buf[:parts * rows_ * ld].view(parts, rows_, ld)
    P = planes(parts_p, n, ldp)
    PT = planes(parts_t, k, ldt) if want_t else None
    L = _lib.lib()
    st = stream_of(w)
    if parts_p == parts_t or not want_t:
        fn = L.pdgn_split_bf16x3 if parts_p == 3 else L.pdgn_split_f16x2
        check(fn(n, k, ptr(w), w.stride(0), ptr(P), ldp, ctypes.c_longlong(n * ldp), ptr(PT), ldt,
                 ctypes.c_longlong(k * ldt if want_t else 0), st), "pdgn_split")
    else:                                                          # one form each: two launches
        for parts, args in ((parts_p, (ptr(P), ldp, ctypes.c_longlong(n * ldp), None, 0, ctypes.c_longlong(0))),
                            (parts_t, (None, 0, ctypes.c_longlong(0), ptr(PT), ldt, ctypes.c_longlong(k * ldt)))):
            fn = L.pdgn_split_bf16x3 if parts == 3 else L.pdgn_split_f16x2
            check(fn(n, k, ptr(w), w.stride(0), *args, st), "pdgn_split")
    return Planes(P, PT, (n, k), parts_p, parts_t)


def _maxima_ok(t):
    return (_lib.gemm_mode() == "x2" and t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and t.dtype == F32 and t.shape[1] % 4 == 0
            and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0)


def operand_maxima(t, rows=True, cols=False):
    """Two-part mode (csrc/gemm_x3.hip, "x2"): the scales of a 2-D fp32 operand are one power of two per ROW of the operand as the
    contraction kernel sees it, derived from that row's largest magnitude.  This is ONE scan (pdgn_absmax_rows_cols) that leaves
    the row maxima (int32 (rows,): bit patterns of |x|) and / or the column maxima ((cols,): what a product that takes the
    operand TRANSPOSED -- a weight gradient -- needs) in tensors of their own, to hand to every contraction the operand feeds
    instead of a scan per call.  Returns rowmax, colmax, or the pair when both are asked for; None in the other modes.
    (Round 5 kept 256 partial maxima per operand in slots of one process-wide ring; the arrays are ordinary tensors now: they
    live as long as their users hold them, on the operand's device -- ADVICE r5.)"""
    if not _maxima_ok(t) or not (rows or cols):
        return (None, None) if (rows and cols) else None
    rm = torch.empty((t.shape[0],), dtype=torch.int32, device=t.device) if rows else None
    cm = torch.empty((t.shape[1],), dtype=torch.int32, device=t.device) if cols else None
    check(_lib.lib().pdgn_absmax_rows_cols(ctypes.c_longlong(t.shape[0]), t.shape[1], ptr(t), t.stride(0), ptr(rm), ptr(cm),
                                            stream_of(t)), "pdgn_absmax_rows_cols")
    return (rm, cm) if (rows and cols) else (rm if rows else cm)


def row_maxima_buffer(rows, device):
    """An int32 (rows,) tensor for a producer kernel that writes its output's row maxima itself (it zero-fills it first)."""
    return torch.empty((rows,), dtype=torch.int32, device=device)


def _hand_maxima(L, ma, mw=None):
    if ma is not None or mw is not None:
        L.pdgn_gemm_set_operand_scales(ptr(ma), ptr(mw))


def _tail_workspace(L, m, n, k, with_stats, device, parts=None):
    """The stream-K tail of the next contraction call without atomics (csrc/gemm_x3.hip): a buffer for its partial tiles, handed to
    the library for that ONE call.  Returned so that the caller keeps it alive until its launch is issued (stream-ordered reuse
    by the caching allocator is safe: the next user runs behind this call on the same stream).  parts: of the pre-split planes the
    call multiplies (two-part planes run on the 256 x 128 tile whatever the launch model picks)."""
    if parts is not None:
        L.pdgn_gemm_nt_ps_workspace_floats.restype = ctypes.c_longlong
        need = L.pdgn_gemm_nt_ps_workspace_floats(ctypes.c_longlong(m), n, k, parts, 1 if with_stats else 0)
    else:
        L.pdgn_gemm_tail_workspace_floats.restype = ctypes.c_longlong
        need = L.pdgn_gemm_tail_workspace_floats(ctypes.c_longlong(m), n, k, 1 if with_stats else 0)
    if need <= 0:
        return None
    ws = torch.empty(need, dtype=F32, device=device)
    check(L.pdgn_gemm_set_tail_workspace(ptr(ws), ctypes.c_longlong(need)), "pdgn_gemm_set_tail_workspace")
    return ws


def _rp_takes(m, n, k, parts, plain):
    """Whether pdgn_gemm_nt_ps(m, n, k) on `parts`-part planes runs on the row-panel kernel (csrc/gemm_rp.hip: short reductions on
    two-part planes, no bias / addend): its BatchNorm partials cover a 256-row panel each."""
    return parts == 2 and plain and _lib.lib().pdgn_gemm_nt_ps_row_panel(ctypes.c_longlong(m), n, k, 2, 1) == 1


def planes_fit(P, m, n, k, with_stats=False, plain=False):
    """Whether planes P can serve the product (m, n, k): three-part planes always; two-part ones (which run on the 256 x 128 tile
    whatever the launch model picks, or -- short reductions without bias / addend: plain -- on the row-panel kernel) unless the
    launch emits BatchNorm partials and neither the row-panel kernel takes it nor the model's pick -- whose geometry the partials'
    consumers were told -- is the 256 x 128 tile."""
    return (P.shape[0] == 3 or not with_stats or _rp_takes(m, n, k, P.shape[0], plain)
            or (_lib.lib().pdgn_gemm_nt_config(ctypes.c_longlong(m), n, k, 1) & 15) == 0)


def gemm_nt_planes(a, P, n, k, bias=None, addend=None, want_stats=False, max_a=None):
    """a (m, k) @ W^T for a weight given as its planes P [parts][n][ld] (Planes.p of W, or Planes.t for the product with W itself:
    then n, k are W^T's; two-part planes: max_a = a's maxima when the caller has them, operand_maxima); everything else as gemm_nt."""
    _slots(a)
    m = a.shape[0]
    if GEMM_LOG is not None:
        GEMM_LOG.append(("nt", m, n, k))
    ap = _pad_cols(a)
    L = _lib.lib()
    out = torch.empty((m, n), dtype=F32, device=a.device)
    part = None
    if want_stats:
        L.pdgn_gemm_nt_ps_stat_rows.restype = ctypes.c_longlong
        part = torch.empty((L.pdgn_gemm_nt_ps_stat_rows(ctypes.c_longlong(m), n, k, P.shape[0], 1 if (bias is None and addend is None) else 0),
                            3 * n), dtype=F32, device=a.device)
    b = bias.detach().contiguous() if bias is not None else None
    if addend is not None:
        addend = _pad_cols(addend)
    ws = _tail_workspace(L, m, n, k, want_stats, a.device, parts=P.shape[0])
    _hand_maxima(L, max_a)
    check(L.pdgn_gemm_nt_ps(ctypes.c_longlong(m), n, k, ptr(ap), ap.stride(0), ptr(P), P.shape[2], ctypes.c_longlong(P.shape[1] * P.shape[2]), P.shape[0],
                            ptr(b), ptr(addend), addend.stride(0) if addend is not None else 0, ptr(out), n, ptr(part), None, 0, 1, 0,
                            None, 0, stream_of(a)), "pdgn_gemm_nt_ps")
    return (out, part) if want_stats else out


def gemm_nt(a, w, bias=None, addend=None, want_stats=False, w_transposed=False, max_a=None):
    """a (m, k) @ w (n, k)^T (+ bias) (+ addend) on pdgn_gemm_nt -- or, with w_transposed, a (m, k) @ w (k, n) on
    pdgn_gemm_nn (the input gradient dy @ W straight from the layer's weight).  Channel counts that are not multiples
    of 4 (the xyz layers: k = 3, the heads' last conv: n = 3) are zero-padded for the launch.  want_stats: also returns
    the BatchNorm partials of the result ((parts, 3n) fp32, block-shifted: per-column sum (x - pv) | sum (x - pv)^2 | pv of
    row blocks of pdgn_gemm_nt_stat_block_rows(m, n, k) rows, pv = the block's first row)."""
    _slots(a)
    m, k = a.shape
    n = w.shape[1] if w_transposed else w.shape[0]
    if GEMM_LOG is not None:
        GEMM_LOG.append(("nn" if w_transposed else "nt", m, n, k))
    ap = _pad_cols(a)
    kp = ap.shape[1]
    np_ = (n + 3) // 4 * 4
    if w_transposed:
        wp = _pad_cols(w)                                          # (k, n) -> (k, np_)
        if kp != k:
            wp = torch.nn.functional.pad(wp, (0, 0, 0, kp - k))
    else:
        wp = _pad_cols(w)
    if np_ != n and not w_transposed:
        wp = torch.nn.functional.pad(wp, (0, 0, 0, np_ - n))
    if np_ != n:
        bias = torch.nn.functional.pad(bias, (0, np_ - n)) if bias is not None else None
        addend = torch.nn.functional.pad(addend, (0, np_ - n)) if addend is not None else None
    if addend is not None:
        addend = _pad_cols(addend)
    L = _lib.lib()
    out = torch.empty((m, np_), dtype=F32, device=a.device)
    part = None
    if want_stats:
        L.pdgn_gemm_nt_stat_rows.restype = ctypes.c_longlong
        part = torch.empty((L.pdgn_gemm_nt_stat_rows(ctypes.c_longlong(m), np_, kp), 3 * np_), dtype=F32, device=a.device)
    b = bias.detach().contiguous() if bias is not None else None
    fn = L.pdgn_gemm_nn if w_transposed else L.pdgn_gemm_nt
    ws = _tail_workspace(L, m, np_, kp, want_stats, a.device)       # (kept alive to the end of this function: the launch is issued by then)
    _hand_maxima(L, max_a)
    check(fn(ctypes.c_longlong(m), np_, kp, ptr(ap), ap.stride(0), ptr(wp), wp.stride(0), ptr(b), ptr(addend),
             addend.stride(0) if addend is not None else 0, ptr(out), np_, ptr(part), stream_of(a)),
          "pdgn_gemm_nn" if w_transposed else "pdgn_gemm_nt")
    if np_ != n:
        out = out[:, :n].contiguous()
    return (out, part) if want_stats else out


def gemm_tn(dy, x, max_dy=None, max_x=None):
    """dy (m, n)^T @ x (m, k) -> (n, k) on pdgn_gemm_tn (reduction over the rows split over workgroups).  max_dy / max_x: the COLUMN
    maxima of dy / x (operand_maxima(..., cols=True)) for a two-part product, or None (the library scans where two parts pay)."""
    _slots(dy)
    m, n = dy.shape
    k = x.shape[1]
    if GEMM_LOG is not None:
        GEMM_LOG.append(("tn", m, n, k))
    dyp, xp = _pad_cols(dy), _pad_cols(x)
    nk = dyp.shape[1] * xp.shape[1]
    # (long reductions -- >= 150 k rows -- also with 16 K .. 64 K outputs: 0.85-0.88x pdgn_gemm_tn's time, r03_gemm_shapes.txt)
    if _TN_BIG and dyp.shape[1] >= 64 and xp.shape[1] >= 64 and (65536 if (m < 150000 or not _lib.matrix_core_mode()) else 16384) <= nk <= _tn_big_max():
        # mid-sized outputs (4 .. 64 tiles of 128 x 128): the stream-K launch of the pdgn_gemm_nt kernel with both operands
        # transposed balances them better than pdgn_gemm_tn's split (measured, tools/gemm_shapes.py: 0.70-0.94x its time);
        # smaller outputs (and, on the fp32 kernels, the two largest ones: conv2's dense half, the per-point GEMM) stay on
        # pdgn_gemm_tn
        L = _lib.lib()
        L.pdgn_gemm_tn_big_workspace_floats.restype = ctypes.c_longlong
        need = L.pdgn_gemm_tn_big_workspace_floats(ctypes.c_longlong(m), dyp.shape[1], xp.shape[1])
        if need > 0:                                               # the k slices' partial tiles: summed in a fixed order, no atomics
            ws = torch.empty(need, dtype=F32, device=dy.device)
            check(L.pdgn_gemm_set_tail_workspace(ptr(ws), ctypes.c_longlong(need)), "pdgn_gemm_set_tail_workspace")
            dwp = torch.empty((dyp.shape[1], xp.shape[1]), dtype=F32, device=dy.device)
        else:
            dwp = _zeros((dyp.shape[1], xp.shape[1]), dy.device)   # a slice of the backward pass's zero arena: no fill launch here
        _hand_maxima(L, max_dy, max_x)
        check(_lib.lib().pdgn_gemm_tn_big(ctypes.c_longlong(m), dyp.shape[1], xp.shape[1], ptr(dyp), dyp.stride(0), ptr(xp),
                                          xp.stride(0), ptr(dwp), 0 if need > 0 else 1, stream_of(dy)), "pdgn_gemm_tn_big")      # (not zero-filled when the workspace form is expected: the atomic form, should the library take it after all, fills it itself)
        return dwp if (dyp.shape[1] == n and xp.shape[1] == k) else dwp[:n, :k]
    if dyp.stride(0) != dyp.shape[1]:
        dyp = dyp.contiguous()
    if xp.stride(0) != xp.shape[1]:
        xp = xp.contiguous()
    dwp = _zeros((dyp.shape[1], xp.shape[1]), dy.device)
    check(_lib.lib().pdgn_gemm_tn(ctypes.c_longlong(m), dyp.shape[1], xp.shape[1], ptr(dyp), ptr(xp), ptr(dwp),
                                  stream_of(dy)), "pdgn_gemm_tn")
    return dwp if (dyp.shape[1] == n and xp.shape[1] == k) else dwp[:n, :k]


def _thin_ok(x, n, k):
    """The layer has <= 4 channels on one side and fits pdgn_thin_nt (csrc/thin.hip): no zero-padded copies."""
    return (k <= 4 and n % 4 == 0 and n <= 1024) or (n <= 4 and k % 4 == 0 and k <= 1024)   # (the backward runs the wide side as n)


def thin_nt(x, w, wrs, wcs, n, bias=None, want_stats=False):
    """x (m, k) times W'^T with W'[j, kk] = w.flat[j * wrs + kk * wcs] (n rows) on pdgn_thin_nt; want_stats as gemm_nt."""
    m, k = x.shape
    if GEMM_LOG is not None:
        GEMM_LOG.append(("thin", m, n, k))
    x = x if (x.stride(1) == 1 and (k <= 4 or (x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0))) else x.contiguous()
    L = _lib.lib()
    out = torch.empty((m, n), dtype=F32, device=x.device)
    part = None
    if want_stats and k <= 4:
        L.pdgn_thin_stat_rows.restype = ctypes.c_longlong
        part = torch.empty((L.pdgn_thin_stat_rows(ctypes.c_longlong(m)), 3 * n), dtype=F32, device=x.device)
    b = bias.detach().contiguous() if bias is not None else None
    check(L.pdgn_thin_nt(ctypes.c_longlong(m), n, k, ptr(x), x.stride(0), ptr(w), wrs, wcs, ptr(b), ptr(out), n, ptr(part),
                         stream_of(x)), "pdgn_thin_nt")
    return (out, part) if want_stats else out


def thin_tn(dy, x, want_db):
    """dW (n, k) = dy (m, n)^T x (m, k) [, db = column sums of dy] for a layer with n <= 4 or k <= 4, on pdgn_thin_tn."""
    m, n = dy.shape
    k = x.shape[1]
    if GEMM_LOG is not None:
        GEMM_LOG.append(("thin_tn", m, n, k))
    dw = _zeros((n, k), dy.device)
    db = _zeros((n,), dy.device) if want_db else None
    L = _lib.lib()
    if k <= 4:                      # A = x (thin), B = dy (wide): O[i = kk, j = n] -> dw[j, i]
        check(L.pdgn_thin_tn(ctypes.c_longlong(m), k, n, ptr(x), x.stride(0), ptr(dy), dy.stride(0), ptr(dw), 1, k, ptr(None),
                             ptr(db), stream_of(dy)), "pdgn_thin_tn")
    else:                           # A = dy (thin), B = x (wide): O[i = n, j = kk] -> dw[i, j]
        check(L.pdgn_thin_tn(ctypes.c_longlong(m), n, k, ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(dw), k, 1, ptr(db),
                             ptr(None), stream_of(dy)), "pdgn_thin_tn")
    return dw, db


def _planes_taken(x, weight, planes, want_stats, plain):
    """Whether LinearCL's forward multiplies against the pre-split planes (the same question stat_block_rows asks)."""
    return (planes is not None and x.is_cuda and x.shape[0] >= _PLANES_MIN_ROWS and planes.shape == tuple(weight.shape)
            and x.shape[1] == weight.shape[1] and x.shape[1] % 4 == 0
            and planes_fit(planes.p, x.shape[0], weight.shape[0], weight.shape[1], want_stats, plain))


class LinearCL(Function):
    """y = x @ W^T (+ b) (+ addend) for point-major rows x (M, C_in): the reference's Conv2d / Conv1d / Linear layers
    (models/PDGNet_v2.py:559-625, 835-862, 886-1014) as row-matrix products on the hand-written MFMA kernels (see above)."""

    @staticmethod
    def forward(ctx, x, weight, bias, addend, want_stats=False, planes=None, x_max=None, x_cmax=None):
        ctx.save_for_backward(x, weight)
        ctx.set_materialize_grads(False)          # no zero-filled "gradient" of the statistics partials (a launch per call)
        ctx.has_bias = bias is not None
        ctx.has_addend = addend is not None
        ctx.planes_t = None
        ctx.max_x = None
        ctx.max_x_cols = x_cmax                                # x's column maxima when its producer computed them (the weight gradient's scale)
        if _planes_taken(x, weight, planes, want_stats, bias is None and addend is None):
            # the weight arrives pre-split (Planes): no split work for it in the kernel, forward and input gradient
            n, k = weight.shape
            ctx.thin = False
            ctx.planes_t = planes.t
            # (a two-part product scales x row by row: its row maxima from the producer, or one scan -- or, on the row-panel kernel,
            # taken inside the launch: a wave holds whole rows)
            xm = None
            if planes.parts_p == 2 and not _rp_takes(x.shape[0], n, k, 2, bias is None and addend is None):
                xm = x_max if x_max is not None else operand_maxima(x)
            ctx.max_x = xm
            if want_stats:
                y, part = gemm_nt_planes(x, planes.p, n, k, bias, addend, want_stats=True, max_a=xm)
                ctx.mark_non_differentiable(part)
                return y, part
            return gemm_nt_planes(x, planes.p, n, k, bias, addend, max_a=xm)
        ctx.thin = bool(x.is_cuda and x.shape[0] >= _OWN_MIN_ROWS and addend is None and weight.is_contiguous()
                        and _thin_ok(x, weight.shape[0], weight.shape[1]))
        if ctx.thin:
            n, k = weight.shape
            if want_stats and k <= 4:
                y, part = thin_nt(x, weight, k, 1, n, bias, want_stats=True)
                ctx.mark_non_differentiable(part)
                return y, part
            y = thin_nt(x, weight, k, 1, n, bias)
            return (y, None) if want_stats else y
        if x.is_cuda and x.shape[0] >= _OWN_MIN_ROWS:
            xm = None                                              # (the library scans both operands itself where two parts pay)
            if want_stats and weight.shape[0] % 4 == 0:
                y, part = gemm_nt(x, weight, bias, addend, want_stats=True, max_a=xm)
                ctx.mark_non_differentiable(part)
                return y, part
            y = gemm_nt(x, weight, bias, addend, max_a=xm)
            return (y, None) if want_stats else y
        y = torch.nn.functional.linear(x, weight, bias)
        y = y + addend if addend is not None else y
        return (y, None) if want_stats else y

    @staticmethod
    def backward(ctx, dy, *unused):
        x, weight = ctx.saved_tensors
        if dy is None:
            return None, None, None, None, None, None, None, None
        carried = take_input_grad(dy)
        if carried is not None:              # BNActMaxPool's backward already carried the gradient through this layer
            return carried[0], carried[1], None, None, None, None, None, None
        if is_placeholder(dy):
            raise RuntimeError("LinearCL.backward: received BNActMaxPool's gradient placeholder without the gradient it stands for "
                               "(another consumer of the layer's output, a hook or a copy sits between the two nodes)")
        zero_db = ctx.has_bias and ctx.needs_input_grad[2] and has_zero_colsum(dy)
        dy = dy.contiguous()
        own = dy.is_cuda and dy.shape[0] >= _OWN_MIN_ROWS
        dx = dw = db = None
        if ctx.thin:
            n, k = weight.shape
            if ctx.needs_input_grad[0]:
                dx = thin_nt(dy, weight, 1, k, k)                  # dX = dY W: W'[j, kk] = weight[kk, j]
            want_db = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                xc = x if (x.stride(1) == 1 and (k <= 4 or (x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0))) else x.contiguous()
                dw, db = thin_tn(dy, xc, want_db and not zero_db)
                if want_db and zero_db:
                    db = _zeros((n,), dy.device)
            elif want_db:
                db = _zeros((n,), dy.device) if zero_db else group_colsum(dy)[0]
            return dx, dw, db, None, None, None, None, None
        # two-part products (mode "x2", where they pay): the input gradient dX = dY W scales dY row by row, the weight gradient
        # dW = dY^T X scales dY and X column by column (its kernel takes both transposed); dy is scanned ONCE for both
        dy_rows = dy_cols = x_cols = None
        if own and _lib.gemm_mode() == "x2" and dy.shape[1] % 4 == 0:
            m_, n_, k_ = dy.shape[0], weight.shape[0], weight.shape[1]
            dy_rows = take_maxima(dy)                              # left behind by the backward that wrote dy (EdgeGatherSum), or None
            dx_two = (ctx.needs_input_grad[0] and ctx.planes_t is not None and ctx.planes_t.shape[0] == 2
                      and planes_fit(ctx.planes_t, m_, k_, n_))
            scan_dy_rows = dx_two and dy_rows is None
            xs = x if x.stride(1) == 1 else None
            dw_two = (ctx.needs_input_grad[1] and xs is not None and _maxima_ok(dy) and _maxima_ok(xs)
                      and two_part(n_, k_, m_, (0 if scan_dy_rows else m_ * n_ * 4) + (0 if ctx.max_x_cols is not None else m_ * k_ * 4)))
            if scan_dy_rows or dw_two:
                r_, c_ = operand_maxima(dy, rows=True, cols=True) if (scan_dy_rows and dw_two) else \
                    ((operand_maxima(dy), None) if scan_dy_rows else (None, operand_maxima(dy, rows=False, cols=True)))
                dy_rows = r_ if scan_dy_rows else dy_rows
                dy_cols = c_
            if dw_two:
                x_cols = ctx.max_x_cols if ctx.max_x_cols is not None else operand_maxima(xs, rows=False, cols=True)
        if ctx.needs_input_grad[0]:
            if ctx.planes_t is not None and own and dy.shape[1] % 4 == 0 and planes_fit(ctx.planes_t, dy.shape[0], weight.shape[1], weight.shape[0]):
                dx = gemm_nt_planes(dy, ctx.planes_t, weight.shape[1], weight.shape[0], max_a=dy_rows)      # dX = dY W = dY (W^T)^T
            else:
                dx = gemm_nt(dy, weight, w_transposed=True, max_a=dy_rows) if own else dy.matmul(weight)      # (where two parts pay the library scans what is not handed in)
        if ctx.needs_input_grad[1]:
            dw = gemm_tn(dy, x, dy_cols, x_cols) if own else dy.t().matmul(x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _zeros((dy.shape[1],), dy.device) if zero_db else group_colsum(dy)[0]
        return dx, dw, db, (dy if ctx.has_addend and ctx.needs_input_grad[3] else None), None, None, None, None


# ---------------------------------------------------------------------------------------------------------------
# Products with a per-sample operand (R <= 64 rows: the batch) on csrc/skinny.hip instead of the BLAS library's skinny solutions.
_FROZEN_HEAD = os.environ.get("PDGN_FROZEN_HEAD", "1") == "1"  # A/B switch: frozen small layers' input gradient in one launch
_SK_NN_MIN_K = int(os.environ.get("PDGN_SKINNY_NN_MINK", "2048"))   # shorter reductions of dx = dy W go to the BLAS library
_SKINNY = os.environ.get("PDGN_SKINNY", "1") == "1"           # A/B switch: 0 = torch's matmul for the 35-row layers


def _sk_ok(a, *others):
    return (_SKINNY and a.is_cuda and a.dim() == 2 and a.shape[0] <= 64 and a.dtype == F32 and
            all(t.is_cuda and t.dtype == F32 and t.stride(-1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 for t in (a,) + others))


def skinny_nt(a, w, bias=None):
    """a (R, K) @ w (N, K)^T (+ bias) for R <= 64; w may be a column slice of a wider weight (its row pitch is used)."""
    R, K = a.shape
    N = w.shape[0]
    if not (_sk_ok(a, w) and K % 4 == 0):
        return torch.addmm(bias, a, w.t()) if bias is not None else a.matmul(w.t())
    out = torch.empty((R, N), dtype=F32, device=a.device)
    b = bias.detach().contiguous() if bias is not None else None
    check(_lib.lib().pdgn_skinny_nt(R, N, K, ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(b), ptr(out), N, stream_of(a)), "pdgn_skinny_nt")
    return out


def skinny_nn(a, w):
    """a (R, K) @ w (K, N) for R <= 64 (the input gradient of a per-sample layer: a = dy, w = the layer's (out, in) weight)."""
    R, K = a.shape
    N = w.shape[1]
    # long reductions only (dconst = dYc WcatC: K = 6432, 12832: 14-22 us against the library's 25-34); for the short ones
    # (dx of the small layers, K <= 1024) the library's 32 x 32 tiles take 4-7 us and this kernel's join + atomics 11-12
    if not (_sk_ok(a, w) and N % 4 == 0 and K >= _SK_NN_MIN_K):
        return a.matmul(w)
    out = _zeros((R, N), a.device)                                 # K slices add into it
    check(_lib.lib().pdgn_skinny_nn(R, N, K, ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(out), N, stream_of(a)), "pdgn_skinny_nn")
    return out


def skinny_nn_masked(a, pre, act, w):
    """(a * act'(pre)) (R, K) @ w (K, N) for R <= 64: the input gradient of a FROZEN Linear + activation in one launch (a = dy, pre = the
    saved pre-activation, act 1 = ReLU / 2 = LeakyReLU(0.01), w = the (out, in) weight)."""
    R, K = a.shape
    N = w.shape[1]
    if not (_sk_ok(a, w) and N % 4 == 0 and K % 4 == 0 and pre.stride(1) == 1 and act in (1, 2)):
        slope = 0.01 if act == 2 else 0.0
        return (a * torch.where(pre > 0, 1.0, slope)).matmul(w) if act else a.matmul(w)
    out = _zeros((R, N), a.device)                                 # K slices add into it
    check(_lib.lib().pdgn_skinny_nn_masked(R, N, K, ptr(a), a.stride(0), ptr(pre), pre.stride(0), act, ptr(w), w.stride(0), ptr(out), N,
                                           stream_of(a)), "pdgn_skinny_nn_masked")
    return out


def skinny_tn(a, b, out=None):
    """a (R, N)^T @ b (R, K) -> (N, K) for R <= 64 (the weight gradient of a per-sample layer); `out`: a (N, K) view to write into
    (its row pitch is used: a column slice of a wider gradient)."""
    R, N = a.shape
    K = b.shape[1]
    if not (_SKINNY and a.is_cuda and R <= 64 and a.stride(1) == 1 and b.stride(1) == 1 and (out is None or out.stride(1) == 1)):
        res = a.t().matmul(b)
        if out is None:
            return res
        out.copy_(res)
        return out
    if out is None:
        out = torch.empty((N, K), dtype=F32, device=a.device)
    check(_lib.lib().pdgn_skinny_tn(R, N, K, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0), stream_of(a)), "pdgn_skinny_tn")
    return out


class SkinnyLinear(Function):
    """y = x W^T for x (R <= 64, K): forward, input and weight gradient on csrc/skinny.hip (one pass over W each)."""

    @staticmethod
    def forward(ctx, x, weight):
        x, weight = x.contiguous(), weight.contiguous()
        ctx.save_for_backward(x, weight)
        return skinny_nt(x, weight)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = skinny_nn(dy, weight) if ctx.needs_input_grad[0] else None
        dw = skinny_tn(dy, x) if ctx.needs_input_grad[1] else None
        return dx, dw


def skinny_linear(x, weight):
    """F.linear(x, weight) for a per-sample x (R <= 64 rows) on the GPU; torch's own elsewhere (host tests)."""
    if x.is_cuda and x.dim() == 2 and x.shape[0] <= 64 and x.shape[1] % 4 == 0 and weight.shape[0] % 4 == 0 and _SKINNY:
        return SkinnyLinear.apply(x, weight)
    return torch.nn.functional.linear(x, weight)


def gemm_nt_ex(a, w, ldw, n, bias=None, row_bias=None, rows_per_group=1, act=0, gate=None, w_transposed=False):
    """pdgn_gemm_nt_ex: act(a (m, k) @ W^T + bias + row_bias[row // rows_per_group]) * lrelu'(gate), W given as a data pointer
    with row pitch ldw ((n, k) rows, or (k, n) with w_transposed): a column slice of a wider weight needs no copy."""
    _slots(a)
    m, k = a.shape
    if GEMM_LOG is not None:
        GEMM_LOG.append(("nn" if w_transposed else "nt", m, n, k))
    out = torch.empty((m, n), dtype=F32, device=a.device)
    b = bias.detach().contiguous() if bias is not None else None
    check(_lib.lib().pdgn_gemm_nt_ex(ctypes.c_longlong(m), n, k, ptr(a), a.stride(0), ptr(w), ldw, ptr(b), None, 0, ptr(out), n, None,
                                     ptr(row_bias), row_bias.stride(0) if row_bias is not None else 0, rows_per_group, act,
                                     ptr(gate), gate.stride(0) if gate is not None else 0, int(w_transposed), stream_of(a)),
          "pdgn_gemm_nt_ex")
    return out


class HeadMLP(Function):
    """mlp1..mlp4 of the generator (models/PDGNet_v2.py:835-862) on point-major rows whose input is cat([g broadcast, x]):
        y1 = LeakyReLU(x W0[:, nc:]^T + (g W0[:, :nc]^T + b0)[sample]),  y2 = LeakyReLU(y1 W2^T + b2),  p = y2 W3^T + b3
    with the per-sample term, both activations and -- in backward -- both activation derivatives in GEMM / thin-layer epilogues
    (pdgn_gemm_nt_ex, pdgn_thin_nt_ex): no elementwise pass over the (rows, 256) / (rows, 64) hidden tensors in either direction
    (they were two adds, two LeakyReLUs and two LeakyReLU adjoints per head and pass: ~1 ms of kernel time and ~50 launches per
    iteration).  x (B*M, Fo), g (B, nc); W0 (256, nc + Fo), W2 (64, 256), W3 (3, 64) as (C_out, C_in) views."""

    @staticmethod
    def forward(ctx, x, g, W0, b0, W2, b2, W3, b3, B):
        x = x.contiguous()
        rows, Fo = x.shape
        nc, M = g.shape[1], rows // B
        W0, W2, W3 = W0.contiguous(), W2.contiguous(), W3.contiguous()
        rb = skinny_nt(g.contiguous(), W0[:, :nc], b0)                            # (B, 256): the per-sample row (csrc/skinny.hip)
        w0x = W0[:, nc:]                                                          # (256, Fo) view, row pitch nc + Fo
        y1 = gemm_nt_ex(x, w0x, W0.stride(0), W0.shape[0], row_bias=rb, rows_per_group=M, act=2)
        y2 = gemm_nt_ex(y1, W2, W2.stride(0), W2.shape[0], bias=b2, act=2)
        p = thin_nt(y2, W3, W3.shape[1], 1, W3.shape[0], b3)
        ctx.save_for_backward(x, g, W0, W2, W3, y1, y2)
        ctx.cfg = (B, M, nc)
        return p

    @staticmethod
    def backward(ctx, dp):
        x, g, W0, W2, W3, y1, y2 = ctx.saved_tensors
        B, M, nc = ctx.cfg
        dp = dp.contiguous()
        rows = x.shape[0]
        L = _lib.lib()
        dW3, db3 = thin_tn(dp, y2, True)
        n3, k3 = W3.shape
        if GEMM_LOG is not None:
            GEMM_LOG.append(("thin", rows, k3, n3))
        dpre2 = torch.empty((rows, k3), dtype=F32, device=x.device)               # (dp W3) * lrelu'(y2)
        check(L.pdgn_thin_nt_ex(ctypes.c_longlong(rows), k3, n3, ptr(dp), dp.stride(0), ptr(W3), 1, k3, None, ptr(dpre2), k3, None,
                                ptr(y2), y2.stride(0), stream_of(x)), "pdgn_thin_nt_ex")
        dW2 = gemm_tn(dpre2, y1)
        db2 = group_colsum(dpre2)[0]
        dpre1 = gemm_nt_ex(dpre2, W2, W2.stride(0), W2.shape[1], gate=y1, w_transposed=True)     # (dpre2 W2) * lrelu'(y1)
        w0x = W0[:, nc:]
        dx = gemm_nt_ex(dpre1, w0x, W0.stride(0), x.shape[1], w_transposed=True) if ctx.needs_input_grad[0] else None
        drb = group_colsum(dpre1, M)                                               # (B, 256): the per-sample term's gradient
        dW0 = torch.empty_like(W0)
        skinny_tn(drb, g.contiguous(), out=dW0[:, :nc])                           # drb^T g straight into its column slice
        dW0[:, nc:].copy_(gemm_tn(dpre1, x))
        dg = skinny_nn(drb, W0[:, :nc]) if ctx.needs_input_grad[1] else None
        return dx, dg, dW0, group_colsum(drb)[0], dW2, db2, dW3, db3, None


_PLANES_MIN_ROWS = 4096        # below this a contraction is a few tiles: the split kernel's launches would cost more than they save


def linear_cl(x2d, weight, bias=None, addend=None, want_stats=None, planes=None, x_max=None, x_cmax=None):
    """Dense layer on point-major rows (see LinearCL); `addend` (M, C_out) is added in the GEMM's epilogue.
    x_max: x2d's partial maxima when its producer computed them (bilateral_weighting's want_max), for a two-part product.
    want_stats (True / False, not None): returns the PAIR (y, partials) -- with True the BatchNorm partial sums of y from
    the GEMM's epilogue, for bn_act / bilateral_weighting's `partials` argument (no statistics pass over y); None when
    this call did not produce them."""
    if want_stats is None:
        return LinearCL.apply(x2d, weight, bias, addend, False, planes, x_max, x_cmax)
    if want_stats:
        y, part = LinearCL.apply(x2d, weight, bias, addend, True, planes, x_max, x_cmax)
        return y, ((part, stat_block_rows(x2d, weight, addend, planes, bias)) if part is not None else None)
    return LinearCL.apply(x2d, weight, bias, addend, False, planes, x_max, x_cmax), None


def stat_block_rows(x2d, weight, addend=None, planes=None, bias=None):
    """Rows of y = x2d weight^T each partial-statistics row of LinearCL's epilogue covers: the same question the launch asked
    (the same deterministic launch model on the same padded sizes), so the block size travels with the partials as a value."""
    L = _lib.lib()
    n, k = weight.shape
    plain = bias is None and addend is None
    if _planes_taken(x2d, weight, planes, True, plain):
        return int(L.pdgn_gemm_nt_ps_stat_block_rows(ctypes.c_longlong(x2d.shape[0]), n, k, planes.p.shape[0], 1 if plain else 0))
    if addend is None and weight.is_contiguous() and _thin_ok(x2d, n, k):
        return int(L.pdgn_thin_stat_block_rows())
    return int(L.pdgn_gemm_nt_stat_block_rows(ctypes.c_longlong(x2d.shape[0]), (n + 3) // 4 * 4, (k + 3) // 4 * 4))


class SoftmaxSlotsPermute(Function):
    """h (M, k, C) -> softmax over the k slots, written as (M, k/2, 2C) with channel 2c+j of row p
    holding slot (k/2)*j + p (models/PDGNet_v2.py:634-641 in one pass)."""

    @staticmethod
    def forward(ctx, h):
        h = h.contiguous()
        m, k, c = h.shape
        w = torch.empty((m, k // 2, 2 * c), dtype=F32, device=h.device)
        check(_lib.lib().pdgn_softmax_slots_permute(ctypes.c_longlong(m), k, c, ptr(h), ptr(w), stream_of(h)),
              "pdgn_softmax_slots_permute")
        ctx.save_for_backward(w)
        ctx.shape = (m, k, c)
        return w

    @staticmethod
    def backward(ctx, dw):
        (w,) = ctx.saved_tensors
        m, k, c = ctx.shape
        dw = dw.contiguous()
        dh = torch.empty((m, k, c), dtype=F32, device=w.device)
        check(_lib.lib().pdgn_softmax_slots_permute_backward(ctypes.c_longlong(m), k, c, ptr(w), ptr(dw), ptr(dh),
                                                             stream_of(w)), "pdgn_softmax_slots_permute_backward")
        return dh


def softmax_slots_permute(h):
    return SoftmaxSlotsPermute.apply(h)


class BNSoftmaxSlotsPermute(Function):
    """softmax_slots_permute(act(BatchNorm(x))) for x (M*k, C) raw conv outputs: the statistics pass, then ONE
    pass that normalises, activates, soft-maxes over the k slots and writes the interleaved weights; the
    activated logits never exist in HBM.  Backward: softmax adjoint, then the BatchNorm+act adjoint on x."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, act, k, pre_bias=None):
        rows, C = x.shape
        m = rows // k
        x = x.contiguous()
        L = _lib.lib()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        stats = _bn_stats(L, x, rows, C, g, b, pre_bias, running_mean, running_var, training, momentum, eps)
        ctx.has_pre_bias = pre_bias is not None
        w = torch.empty((m, k // 2, 2 * C), dtype=F32, device=x.device)
        check(L.pdgn_bn_softmax_slots_permute(ctypes.c_longlong(m), k, C, act, ptr(x), ptr(stats), ptr(w),
                                              stream_of(x)), "pdgn_bn_softmax_slots_permute")
        ctx.save_for_backward(x, stats, w)
        ctx.cfg = (rows, C, act, bool(training), k)
        return w

    @staticmethod
    def backward(ctx, dw):
        x, stats, w = ctx.saved_tensors
        rows, C, act, training, k = ctx.cfg
        L = _lib.lib()
        dw = dw.contiguous()
        dh = torch.empty((rows, C), dtype=F32, device=x.device)
        check(L.pdgn_softmax_slots_permute_backward(ctypes.c_longlong(rows // k), k, C, ptr(w), ptr(dw), ptr(dh),
                                                    stream_of(w)), "pdgn_softmax_slots_permute_backward")
        scratch = torch.empty(_scratch_floats(L, rows, C), dtype=F32, device=x.device)
        bs = torch.empty(2 * C, dtype=F32, device=x.device)
        dx = torch.empty_like(x)
        check(L.pdgn_bn_act_backward(ctypes.c_longlong(rows), C, act, int(training), ptr(x), ptr(dh), ptr(None),
                                     ptr(stats), ptr(scratch), ptr(bs), ptr(dx), ptr(None), 0, stream_of(x)),
              "pdgn_bn_act_backward")
        if training:
            mark_zero_colsum(dx)
        return dx, bs[C:], bs[:C], None, None, None, None, None, None, None, _pre_bias_grad(ctx.has_pre_bias, C, x.device)


def bn_softmax_slots_permute(x2d, bn, training, k, act="leaky_relu", pre_bias=None):
    """x2d (M*k, C) -> (M, k/2, 2C): nn.BatchNorm2d `bn` + `act` + softmax over the k slots + interleave."""
    x2d, pre_bias = _fold_pre_bias(x2d, pre_bias, training)
    if x2d.shape[1] % 4:
        return softmax_slots_permute(bn_act(x2d, bn, training, act=act, pre_bias=pre_bias).view(-1, k, x2d.shape[1]))
    if training and bn.track_running_stats:
        _PENDING_COUNTS[bn.num_batches_tracked] = _PENDING_COUNTS.get(bn.num_batches_tracked, 0) + 1
    return BNSoftmaxSlotsPermute.apply(x2d, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum,
                                       bn.eps, ACT[act], k, pre_bias)


class BilateralWeighting(Function):
    """y = act(BN_u(u)) * softmax_slots_permute(act(BN_x(x)))  -- the bilateral weighting of a deconvolution block
    (models/PDGNet_v2.py:623-642) with both BatchNorms, both activations, the slot softmax, the channel interleave and
    the product in one pass over x (M*k, C) and u (M*k/2, 2C).  Saved for backward: x, u and the two statistics rows;
    the softmax weights w are recomputed from x by the fused adjoint (k = 4, 10) and never reach HBM."""

    @staticmethod
    def forward(ctx, x, u, gx, bx, rmx, rvx, pbx, gu, bu, rmu, rvu, pbu, training, momentum_x, eps_x, momentum_u, eps_u,
                act, k, partials_u=None, partials_x=None, want_max=False):
        rows, C = x.shape
        m = rows // k
        x, u = x.contiguous(), u.contiguous()
        L = _lib.lib()
        stats_x = _bn_stats(L, x, rows, C, gx.detach().contiguous(), bx.detach().contiguous(), pbx, rmx, rvx, training,
                            momentum_x, eps_x, partials_x)
        stats_u = _bn_stats(L, u, u.shape[0], 2 * C, gu.detach().contiguous(), bu.detach().contiguous(), pbu, rmu, rvu,
                            training, momentum_u, eps_u, partials_u)
        # the fused adjoint (k = 4, 10) recomputes w from x; only the generic chain of separate adjoint kernels reads it back
        need_w = any(ctx.needs_input_grad) and k not in (4, 10)
        w = torch.empty((m, k // 2, 2 * C), dtype=F32, device=x.device) if need_w else None
        y = torch.empty_like(u)
        # want_max: y's partial maxima come out of the same pass (the two-part contraction that takes y would scan it otherwise)
        ymax = row_maxima_buffer(m, x.device) if want_max else None     # y as the (m, k C) operand of conv2's dense half: its ROW maxima
        # ... and, when a backward pass will follow, an upper BOUND of its column maxima (conv2's weight gradient takes y
        # transposed and scales it column by column): y = act(gamma xhat + beta) * w with softmax weights w <= 1 and, under batch
        # statistics, |xhat| <= sqrt(n - 1) for n samples -- so |y[:, (p, c)]| <= |gamma_c| sqrt(n - 1) + |beta_c|, from the 2C
        # parameters alone.  A bound 2^b above the true maximum costs b of the 16 binades over which a value keeps its full 22
        # bits (here b ~ 6: sqrt(n) against the ~5 sigma a column really reaches); the product's error stays far below the fp32
        # accumulation's (gemm_x3.hip).  Measured against the alternatives: a scan of y is 176 us at stage 4, exact maxima from
        # this kernel (threads walking several points) made it 130 us slower per launch.
        # (the bound is written by the same launch: computed with tensor expressions it was four launches per block and pass on the
        # issuing stream)
        ycmax, g_u, b_u, bound = None, None, None, 0.0
        if want_max and training and any(ctx.needs_input_grad):
            ycmax = torch.empty((k // 2 * 2 * C,), dtype=torch.int32, device=x.device)
            g_u, b_u = gu.detach().contiguous(), bu.detach().contiguous()
            bound = max(float(u.shape[0]) - 1.0, 1.0) ** 0.5
        check(L.pdgn_bn_softmax_slots_permute_mul(ctypes.c_longlong(m), k, C, act, ptr(x), ptr(stats_x), act, ptr(u),
                                                  ptr(stats_u), ptr(w), ptr(y), ptr(ymax), ptr(g_u), ptr(b_u), ctypes.c_float(bound),
                                                  ptr(ycmax), stream_of(x)),
              "pdgn_bn_softmax_slots_permute_mul")
        ctx.save_for_backward(x, u, w, stats_x, stats_u)
        ctx.cfg = (rows, C, act, bool(training), k, pbx is not None, pbu is not None)
        if want_max:
            ctx.mark_non_differentiable(ymax)
            if ycmax is not None:
                ctx.mark_non_differentiable(ycmax)
                return y, ymax, ycmax
            return y, ymax, None
        return y

    @staticmethod
    def backward(ctx, dy, *unused):
        x, u, w, stats_x, stats_u = ctx.saved_tensors
        rows, C, act, training, k, has_pbx, has_pbu = ctx.cfg
        L = _lib.lib()
        dy = dy.contiguous()
        rows_u, Cu = u.shape
        if k in (4, 10):
            # both BatchNorm adjoints and the softmax adjoint in two passes over (x, u, w, dy): dW and dh stay in registers
            L.pdgn_bilateral_scratch_floats.restype = ctypes.c_longlong
            scr = torch.empty(L.pdgn_bilateral_scratch_floats(ctypes.c_longlong(rows // k), k, C), dtype=F32, device=x.device)
            bsx = torch.empty(2 * C, dtype=F32, device=x.device)
            bsu = torch.empty(2 * Cu, dtype=F32, device=x.device)
            dx, du = torch.empty_like(x), torch.empty_like(u)
            check(L.pdgn_bilateral_weighting_backward(ctypes.c_longlong(rows // k), k, C, act, int(training), ptr(x),
                                                      ptr(stats_x), ptr(u), ptr(stats_u), ptr(w), ptr(dy), ptr(scr), ptr(bsx),
                                                      ptr(bsu), ptr(dx), ptr(du), stream_of(x)),
                  "pdgn_bilateral_weighting_backward")
        else:
            # y = act(BN(u)) * w: adjoint wrt u, the BatchNorm parameters and w
            scr = torch.empty(_scratch_floats(L, rows_u, Cu), dtype=F32, device=x.device)
            bsu = torch.empty(2 * Cu, dtype=F32, device=x.device)
            du, dw = torch.empty_like(u), torch.empty_like(u)
            check(L.pdgn_bn_act_backward(ctypes.c_longlong(rows_u), Cu, act, int(training), ptr(u), ptr(dy), ptr(w),
                                         ptr(stats_u), ptr(scr), ptr(bsu), ptr(du), ptr(dw), 0, stream_of(x)),
                  "pdgn_bn_act_backward")
            # w = softmax_slots_permute(act(BN(x)))
            dh = torch.empty((rows, C), dtype=F32, device=x.device)
            check(L.pdgn_softmax_slots_permute_backward(ctypes.c_longlong(rows // k), k, C, ptr(w), ptr(dw), ptr(dh),
                                                        stream_of(x)), "pdgn_softmax_slots_permute_backward")
            scr = torch.empty(_scratch_floats(L, rows, C), dtype=F32, device=x.device)
            bsx = torch.empty(2 * C, dtype=F32, device=x.device)
            dx = torch.empty_like(x)
            check(L.pdgn_bn_act_backward(ctypes.c_longlong(rows), C, act, int(training), ptr(x), ptr(dh), ptr(None),
                                         ptr(stats_x), ptr(scr), ptr(bsx), ptr(dx), ptr(None), 0, stream_of(x)),
                  "pdgn_bn_act_backward")
        if training:
            mark_zero_colsum(du)
            mark_zero_colsum(dx)
        return (dx, du, bsx[C:], bsx[:C], None, None, _pre_bias_grad(has_pbx, C, x.device), bsu[Cu:], bsu[:Cu], None, None,
                _pre_bias_grad(has_pbu, Cu, x.device), None, None, None, None, None, None, None, None, None, None)


def bilateral_weighting(x2d, bn_x, u2d, bn_u, training, k, act="leaky_relu", pre_bias_x=None, pre_bias_u=None,
                        partials_u=None, partials_x=None, want_max=False):
    """x2d (M*k, C) raw conv_all.3 output, u2d (M*k/2, 2C) raw inte_conv_hk output ->
    act(bn_u(u2d)) * softmax_slots_permute(act(bn_x(x2d))), shape of u2d.  want_max: returns (y, row maxima, column maxima) of y as
    the (M, k C) operand of conv2's dense half (int32 bit patterns: linear_cl's x_max / x_cmax -- the two-part forward product
    scales y row by row, its weight gradient column by column); either is None when this path did not produce it."""
    x2d, pre_bias_x = _fold_pre_bias(x2d, pre_bias_x, training)
    u2d, pre_bias_u = _fold_pre_bias(u2d, pre_bias_u, training)
    if x2d.shape[1] % 4 or k > 16:           # (the fused adjoint holds all k slots of a channel pair in registers: k <= 16)
        w = bn_softmax_slots_permute(x2d, bn_x, training, k, act=act, pre_bias=pre_bias_x)
        y = bn_act(u2d, bn_u, training, act=act, mul=w.view(u2d.shape), pre_bias=pre_bias_u, partials=partials_u)
        return (y, None, None) if want_max else y
    if training:
        for bn in (bn_x, bn_u):
            if bn.track_running_stats:
                _PENDING_COUNTS[bn.num_batches_tracked] = _PENDING_COUNTS.get(bn.num_batches_tracked, 0) + 1
    return BilateralWeighting.apply(x2d, u2d, bn_x.weight, bn_x.bias, bn_x.running_mean, bn_x.running_var, pre_bias_x,
                                    bn_u.weight, bn_u.bias, bn_u.running_mean, bn_u.running_var, pre_bias_u, training,
                                    bn_x.momentum, bn_x.eps, bn_u.momentum, bn_u.eps, ACT[act], k, partials_u, partials_x, want_max)


class SmallLinearBNAct(Function):
    """act(BatchNorm1d(x W^T + b)) for a handful of rows (x (R<=64, K<=1024)) in one launch forward and one backward
    (+ one small GEMM for dx): the generator's per-sample layers (csrc/small_mlp.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, bn_mode, momentum, eps, act):
        x, weight = x.contiguous(), weight.contiguous()
        R, K = x.shape
        N = weight.shape[0]
        dev = x.device
        need = any(ctx.needs_input_grad)
        y = torch.empty((R, N), dtype=F32, device=dev)
        pre = torch.empty((R, N), dtype=F32, device=dev) if need else None
        stat = torch.empty(2 * N, dtype=F32, device=dev) if (need and bn_mode) else None
        g = gamma.detach().contiguous() if gamma is not None else None
        b = beta.detach().contiguous() if beta is not None else None
        bi = bias.detach().contiguous() if bias is not None else None
        check(_lib.lib().pdgn_small_mlp_forward(R, K, N, act, bn_mode, ctypes.c_float(eps), ctypes.c_float(momentum), ptr(x),
                                                ptr(weight), ptr(bi), ptr(g), ptr(b), ptr(running_mean), ptr(running_var),
                                                ptr(y), ptr(pre), ptr(stat), stream_of(x)), "pdgn_small_mlp_forward")
        ctx.save_for_backward(x, weight, pre, stat, g, b)
        ctx.cfg = (R, K, N, act, bn_mode, bias is not None, gamma is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, pre, stat, g, b = ctx.saved_tensors
        R, K, N, act, bn_mode, has_bias, has_bn = ctx.cfg
        dy = dy.contiguous()
        dev = dy.device
        if not has_bn and not any(ctx.needs_input_grad[1:3]) and _FROZEN_HEAD:
            # frozen Linear (+ activation): only the input gradient is wanted -- one launch (no dpre, no second product)
            if not ctx.needs_input_grad[0]:
                return (None,) * 11
            dx = skinny_nn_masked(dy, pre, act, weight) if act else dy.matmul(weight)
            return (dx,) + (None,) * 10
        dpre = torch.empty((R, N), dtype=F32, device=dev)
        dW = torch.empty((N, K), dtype=F32, device=dev) if ctx.needs_input_grad[1] else None
        dbias = torch.empty(N, dtype=F32, device=dev) if has_bias and ctx.needs_input_grad[2] else None
        dg = torch.empty(N, dtype=F32, device=dev) if has_bn and bn_mode else None
        db = torch.empty(N, dtype=F32, device=dev) if has_bn and bn_mode else None
        check(_lib.lib().pdgn_small_mlp_backward(R, K, N, act, bn_mode, ptr(x), ptr(dy), ptr(pre), ptr(stat), ptr(g), ptr(b),
                                                 ptr(dpre), ptr(dg), ptr(db), ptr(dbias), ptr(dW), stream_of(dy)),
              "pdgn_small_mlp_backward")
        dx = skinny_nn(dpre, weight) if ctx.needs_input_grad[0] else None
        return dx, dW, dbias, dg, db, None, None, None, None, None, None


class SkinnyLinearAct(Function):
    """act(x W^T + b) for a handful of rows without BatchNorm (the discriminators' heads) on csrc/skinny.hip: one matrix-instruction
    launch forward; backward = one launch when the layer is frozen (input gradient with act' applied on load), else
    pdgn_small_mlp_backward (dpre, dW, db) + the dx product.  The activation's derivative is read off y (same sign as the pre-activation)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = x.contiguous()
        R, K = x.shape
        N = weight.shape[0]
        y = torch.empty((R, N), dtype=F32, device=x.device)
        b = bias.detach().contiguous() if bias is not None else None
        check(_lib.lib().pdgn_skinny_nt_act(R, N, K, ptr(x), x.stride(0), ptr(weight), weight.stride(0), ptr(b), ptr(y), N, act,
                                            stream_of(x)), "pdgn_skinny_nt_act")
        ctx.save_for_backward(x, weight, y)
        ctx.cfg = (R, K, N, act, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        R, K, N, act, has_bias = ctx.cfg
        dy = dy.contiguous()
        if not any(ctx.needs_input_grad[1:3]):
            if not ctx.needs_input_grad[0]:
                return None, None, None, None
            return (skinny_nn_masked(dy, y, act, weight) if act else dy.matmul(weight)), None, None, None
        dev = dy.device
        dpre = torch.empty((R, N), dtype=F32, device=dev)
        dW = torch.empty((N, K), dtype=F32, device=dev) if ctx.needs_input_grad[1] else None
        dbias = torch.empty(N, dtype=F32, device=dev) if has_bias and ctx.needs_input_grad[2] else None
        check(_lib.lib().pdgn_small_mlp_backward(R, K, N, act, 0, ptr(x), ptr(dy), ptr(y), None, None, None, ptr(dpre), None, None,
                                                 ptr(dbias), ptr(dW), stream_of(dy)), "pdgn_small_mlp_backward")
        dx = skinny_nn(dpre, weight) if ctx.needs_input_grad[0] else None
        return dx, dW, dbias, None


_SKINNY_HEAD = os.environ.get("PDGN_SKINNY_HEAD", "1") == "1"    # A/B switch


def small_sequential(seq, x, training):
    """nn.Sequential of (Linear [, BatchNorm1d] [, LeakyReLU | ReLU]) groups applied to x (R, K): on a GPU, with at
    most 64 rows, every group is one fused launch (SmallLinearBNAct); otherwise the modules run as they are."""
    mods = list(seq)
    fusable = x.is_cuda and x.dim() == 2 and x.shape[0] <= 64 and all(
        isinstance(m, (torch.nn.Linear, torch.nn.BatchNorm1d, torch.nn.LeakyReLU, torch.nn.ReLU)) for m in mods)
    if fusable:
        fusable = all(m.in_features <= 1024 and x.shape[0] * (m.in_features + 4) * 4 + 1024 <= 160 * 1024
                      for m in mods if isinstance(m, torch.nn.Linear))
    if not fusable:
        return seq(x)
    i = 0
    while i < len(mods):
        lin = mods[i]
        if not isinstance(lin, torch.nn.Linear):
            return seq(x)                                     # unexpected layout: leave it to torch
        i += 1
        bn = None
        if i < len(mods) and isinstance(mods[i], torch.nn.BatchNorm1d):
            bn = mods[i]
            i += 1
        act = 0
        if i < len(mods) and isinstance(mods[i], torch.nn.LeakyReLU):
            if mods[i].negative_slope != 0.01:
                return seq(x)
            act = 2
            i += 1
        elif i < len(mods) and isinstance(mods[i], torch.nn.ReLU):
            act = 1
            i += 1
        if bn is not None:
            use_batch = training or not bn.track_running_stats
            if training and bn.track_running_stats:
                _PENDING_COUNTS[bn.num_batches_tracked] = _PENDING_COUNTS.get(bn.num_batches_tracked, 0) + 1
            x = SmallLinearBNAct.apply(x, lin.weight, lin.bias, bn.weight, bn.bias,
                                       bn.running_mean if bn.track_running_stats else None,
                                       bn.running_var if bn.track_running_stats else None, 1 if use_batch else 2,
                                       bn.momentum if bn.momentum is not None else 0.1, bn.eps, act)
        elif _SKINNY_HEAD and _sk_ok(x, lin.weight) and lin.in_features % 4 == 0:
            x = SkinnyLinearAct.apply(x, lin.weight, lin.bias, act)
        else:
            x = SmallLinearBNAct.apply(x, lin.weight, lin.bias, None, None, None, None, 0, 0.0, 0.0, act)
    return x


class BNActMaxPool(Function):
    """(B*N, C) rows -> (B, C): max over the N points of each sample of act(BatchNorm(x)) -- the
    BatchNorm1d + LeakyReLU + MaxPool1d tail of the discriminators without writing the activated
    tensor; the backward is one streaming pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, act, B, N, pre_bias=None,
                partials=None, dense=None):
        x = x.contiguous()
        rows, C = x.shape
        dev = x.device
        L = _lib.lib()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        ctx.has_pre_bias = pre_bias is not None
        ymax = torch.empty((B, C), dtype=F32, device=dev)
        yarg = torch.empty((B, C), dtype=torch.int32, device=dev)
        if training and partials is None and _STATS_MAX:
            # no statistics from the producer: ONE pass over x for the statistics and the extremes (the sign of the scale picks later)
            stats = torch.empty(4 * C, dtype=F32, device=dev)
            pb = pre_bias.detach().contiguous() if pre_bias is not None else None
            L.pdgn_bn_stats_maxpool_scratch_floats.restype = ctypes.c_longlong
            scr = torch.empty(L.pdgn_bn_stats_maxpool_scratch_floats(B, C), dtype=F32, device=dev)
            check(L.pdgn_bn_stats_act_maxpool(B, N, C, act, ctypes.c_float(eps), ctypes.c_float(momentum), ptr(x), ptr(g), ptr(b), ptr(pb),
                                              ptr(running_mean), ptr(running_var), ptr(scr), ptr(stats), ptr(ymax), ptr(yarg),
                                              stream_of(x)), "pdgn_bn_stats_act_maxpool")
        else:
            stats = _bn_stats(L, x, rows, C, g, b, pre_bias, running_mean, running_var, training, momentum, eps, partials)
            L.pdgn_bn_maxpool_scratch_floats.restype = ctypes.c_longlong
            scr = torch.empty(L.pdgn_bn_maxpool_scratch_floats(B, C), dtype=F32, device=dev)
            check(L.pdgn_bn_act_maxpool(B, N, C, act, ptr(x), ptr(stats), ptr(scr), ptr(ymax), ptr(yarg), stream_of(x)),
                  "pdgn_bn_act_maxpool")
        ctx.save_for_backward(x, stats, yarg)
        ctx.cfg = (B, N, C, act, bool(training))
        # dense layer in front known: the backward goes straight to that layer's input (and weight) gradient
        ctx.dense = dense if (dense is not None and _CLOSED_TAIL and training
                              and not (pre_bias is not None and pre_bias.requires_grad and not _CLOSED_DW)
                              and _dense_input_ok(dense, x, rows, C, N)) else None
        return ymax

    @staticmethod
    def backward(ctx, dout):
        x, stats, yarg = ctx.saved_tensors
        B, N, C, act, training = ctx.cfg
        dout = dout.contiguous()
        if ctx.dense is not None:
            h, w = ctx.dense.h, ctx.dense.w
            K = w.shape[1]
            L = _lib.lib()
            want_w, want_bn = w.requires_grad, any(ctx.needs_input_grad[1:3])
            L.pdgn_dense_bn_maxpool_backward_scratch.restype = ctypes.c_longlong
            scr = torch.empty(L.pdgn_dense_bn_maxpool_backward_scratch(B, C, K), dtype=F32, device=x.device)
            dh = torch.empty((B * N, K), dtype=F32, device=x.device) if h.requires_grad else None
            dw = torch.empty((C, K), dtype=F32, device=x.device) if want_w else None
            bs = torch.empty(2 * C, dtype=F32, device=x.device) if want_bn else None
            check(L.pdgn_dense_bn_maxpool_backward(B, N, C, K, act, ptr(x), ptr(dout), ptr(yarg), ptr(stats), ptr(h), h.stride(0),
                                                   ptr(w), w.stride(0), ptr(scr), ptr(dh), ptr(dw), K, ptr(bs), stream_of(x)),
                  "pdgn_dense_bn_maxpool_backward")
            tok = _placeholder_with_input_grad(B * N, C, dh, dw, x.device)
            return (tok, bs[C:] if want_bn else None, bs[:C] if want_bn else None, None, None, None, None, None, None, None, None,
                    _pre_bias_grad(ctx.has_pre_bias, C, x.device), None, None)
        scr = torch.empty(B * C + 2 * C, dtype=F32, device=x.device)
        bs = torch.empty(2 * C, dtype=F32, device=x.device)
        dx = torch.empty_like(x)
        check(_lib.lib().pdgn_bn_act_maxpool_backward(B, N, C, act, int(training), ptr(x), ptr(dout), ptr(yarg),
                                                      ptr(stats), ptr(scr), ptr(bs), ptr(dx), stream_of(x)),
              "pdgn_bn_act_maxpool_backward")
        if training:
            mark_zero_colsum(dx)
        return (dx, bs[C:], bs[:C], None, None, None, None, None, None, None, None,
                _pre_bias_grad(ctx.has_pre_bias, C, x.device), None, None)


def _is_plain_linear_output(y2d, dense):
    """y2d is the direct output of LinearCL(dense.h, dense.w) without bias or addend (or needs no gradient at all)."""
    fn = y2d.grad_fn
    if fn is None:
        return not y2d.requires_grad
    return (type(fn).__name__ == "LinearCLBackward" and not getattr(fn, "has_bias", True) and not getattr(fn, "has_addend", True)
            and len(fn.saved_tensors) == 2 and fn.saved_tensors[1].data_ptr() == dense.w.data_ptr()
            and fn.saved_tensors[0].data_ptr() == dense.h.data_ptr())


def _dense_input_ok(dense, x, rows, C, N):
    h, w = dense.h, dense.w
    return (x.is_cuda and h.dim() == 2 and w.dim() == 2 and h.shape[0] == rows and w.shape[0] == C and h.shape[1] == w.shape[1]
            and w.shape[1] % 4 == 0 and w.shape[1] <= 256 and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0 and h.dtype == F32 and w.dtype == F32 and h.stride(1) == 1 and w.stride(1) == 1
            and (h.requires_grad or w.requires_grad) and rows >= _OWN_MIN_ROWS
            and (not w.requires_grad or (_CLOSED_DW and w.shape[1] >= 16 and (w.shape[1] & (w.shape[1] - 1)) == 0
                                         and h.stride(0) % 4 == 0 and h.data_ptr() % 16 == 0))
            and N <= 65535 and 2 * (N + min(N, C) + C) + 1040 + 4 * C <= 65536)


def bn_act_maxpool(x2d, bn, training, B, N, act="leaky_relu", pre_bias=None, partials=None, dense=None):
    """max over the N points of every sample of act(BN(x2d)); x2d (B*N, C) -> (B, C).  dense = DenseInput(h, W) when x2d is
    linear_cl(h, W) without bias / addend: with W and the BatchNorm frozen the backward then skips the dense gradient."""
    y2d = x2d
    x2d, pre_bias = _fold_pre_bias(x2d, pre_bias, training)
    if dense is not None and not (x2d is y2d and _is_plain_linear_output(y2d, dense)):
        dense = None                         # the closed tail needs x2d to BE h W^T: the unmodified output of that LinearCL
    if x2d.shape[1] % 4:
        return bn_act(x2d, bn, training, act=act, pre_bias=pre_bias).view(B, N, -1).max(dim=1)[0]
    if training and bn.track_running_stats:
        _PENDING_COUNTS[bn.num_batches_tracked] = _PENDING_COUNTS.get(bn.num_batches_tracked, 0) + 1
    return BNActMaxPool.apply(x2d, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps,
                              ACT[act], B, N, pre_bias, partials, dense)


class PointMax(Function):
    """nn.MaxPool2d((1, N)) over the points of a point-major (B, N, C) tensor (models/PDGNet_v2.py:699/:736/:777/:810):
    values (B, C); the adjoint writes the dense gradient in one pass from the saved argmax (csrc/pointmax.hip)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        require(x, "x", F32, 3)
        b, n, c = x.shape
        L = _lib.lib()
        L.pdgn_point_max_scratch.restype = ctypes.c_longlong
        ns = int(L.pdgn_point_max_scratch(b, n, c))
        sval = torch.empty(ns, dtype=F32, device=x.device)
        sarg = torch.empty(ns, dtype=torch.int32, device=x.device)
        out = torch.empty((b, c), dtype=F32, device=x.device)
        arg = torch.empty((b, c), dtype=torch.int32, device=x.device)
        check(L.pdgn_point_max(b, n, c, ptr(x), ptr(sval), ptr(sarg), ptr(out), ptr(arg), stream_of(x)), "pdgn_point_max")
        ctx.save_for_backward(arg)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        arg, = ctx.saved_tensors
        b, c = arg.shape
        g = g.contiguous()
        if c % 4:
            dx = torch.zeros((b, ctx.n, c), dtype=F32, device=g.device)
            return dx.scatter_(1, arg.long().unsqueeze(1), g.unsqueeze(1))
        dx = torch.empty((b, ctx.n, c), dtype=F32, device=g.device)
        check(_lib.lib().pdgn_point_max_backward(b, ctx.n, c, ptr(g), ptr(arg), ptr(dx), stream_of(g)),
              "pdgn_point_max_backward")
        return dx


def point_max(x):
    """Max over the points (dim 1) of (B, N, C): the HIP kernel pair on a GPU, torch on the CPU (host tests only)."""
    return PointMax.apply(x) if x.is_cuda else x.max(dim=1)[0]
