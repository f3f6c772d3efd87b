"""Live roofline measurement of the hand-written kernels at the launch shapes of one G+D step.

Each entry launches ONE kernel of libpdgn_hip.so through the C ABI on torch's current stream
and times it with HIP events recorded on that same stream (torch.cuda.Event == hipEvent on
ROCm).  `achieved` = algorithmic bytes (or flops) per launch / average launch duration; the
per-unit figures are stated in DESIGN.md section 4.  `traffic` (HBM bytes per launch from the
rocprofv3 PMC counters FETCH_SIZE / WRITE_SIZE, collected in separate passes and corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) is read from profiles/traffic.json, which
tools/pmc_roofline.py writes from the rocprofv3 --pmc passes of tools/run_pmc_roofline.sh.
"""
import ctypes
import json
import os

import torch

from . import _lib
from ._lib import check, ptr, stream_of

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA peak
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: bf16 MFMA, dense
X3_PRODUCTS = 6                # bf16 MFMA products per fp32 product in gemm_x3.hip (csrc/gemm_x3.hip header), mode "x3"
X2_PRODUCTS = 3                # fp16 MFMA products per fp32 product, mode "x2" (two scaled fp16 parts; fp16 and bf16 share the matrix peak)


def products(mode=None):
    """Matrix-core products the contraction kernel issues per fp32 product in `mode` (default: the mode in force); 0: fp32 instructions."""
    return {"x3": X3_PRODUCTS, "x2": X2_PRODUCTS}.get(mode or _lib.gemm_mode(), 0)
_TRAFFIC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")


def _time_us(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def gemm_mode():
    """"x3" (fp32 operands split into three bf16 parts, products on the bf16 matrix cores: csrc/gemm_x3.hip, the default) or
    "fp32" (PDGN_GEMM=fp32: the fp32 matrix instructions, csrc/gemm_nt.hip) -- what pdgn_gemm_nt / nn / tn_big launch."""
    return _lib.gemm_mode()


def _two_part(m, n, k, scan_bytes):
    from .fused import two_part
    return two_part(m, n, k, scan_bytes)


def _entry(name, bound, work, us, x3=False, two=False, **extra):
    nprod = (X2_PRODUCTS if two else X3_PRODUCTS) if x3 else 0
    if bound == "hbm":
        ach, peak, unit = work / us / 1e3, HBM_PEAK_GBS, "GB/s"
        key = "algorithmic_bytes_per_launch"
    else:
        # algorithmic (fp32) flops per second.  The x3 kernel issues X3_PRODUCTS bf16 MFMA flops per algorithmic flop, so its
        # matrix-core ceiling in algorithmic flops is the bf16 peak / X3_PRODUCTS: frac = executed bf16 MFMA rate / bf16 peak.
        ach, unit = work / us / 1e6, "TFLOP/s"
        peak = MFMA_BF16_PEAK_TFLOPS / nprod if x3 else MFMA_F32_PEAK_TFLOPS
        key = "algorithmic_flops_per_launch"
    d = {"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
         "traffic": None, "us_per_launch": us, key: work}
    if bound == "mfma":
        d["mfma"] = ({"instruction": "v_mfma_f32_32x32x16_bf16" if nprod == X3_PRODUCTS else "v_mfma_f32_32x32x16_f16",
                      "products_per_fp32_product": nprod,
                      "executed_tflops": ach * nprod, "instruction_peak_tflops": MFMA_BF16_PEAK_TFLOPS,
                      "fp32_instruction_peak_tflops": MFMA_F32_PEAK_TFLOPS} if x3 else
                     {"instruction": "v_mfma_f32_16x16x4_f32", "instruction_peak_tflops": MFMA_F32_PEAK_TFLOPS})
    d.update(extra)
    return d


def gemm_accuracy(device, M=20000, N=512, K=2560):
    """Largest error of pdgn_gemm_nt against fp64, relative to sum_k |a| |w|, in each arithmetic mode on the same operands (x2: three
    fp16 MFMA products of two scaled parts -- the shape is one the two-part form takes; x3: six bf16 products of three parts; fp32:
    the fp32 matrix instructions; selected per process, _lib.set_gemm_mode): the live form of
    tests/test_gpu_deconv.py::test_gemm_x3_is_as_accurate_as_the_fp32_matrix_instructions."""
    g = torch.Generator(device=device).manual_seed(1234)
    a = torch.randn(M, K, device=device, generator=g) * (torch.rand(M, 1, device=device, generator=g) * 3)
    w = torch.randn(N, K, device=device, generator=g)
    ref = a.double() @ w.double().t()
    mag = a.double().abs() @ w.double().abs().t()
    c = torch.empty(M, N, device=device)
    L = _lib.lib()
    out, saved = {"shape": [M, N, K], "relative_to": "sum_k |a| |w|"}, _lib.gemm_mode()
    # the same with the ROWS of both operands spanning 2^-30 .. 1 of the operand's maximum (points with small activations / small
    # gradients): every element against its own sum_k |a||w| -- a component-wise bound; the two-part form scales each row by its own
    # power of two (round 6), so no row hides behind the operand's norm (tests: test_two_part_rows_spanning_thirty_binades)
    ea = -torch.floor(torch.rand(M, 1, device=device, generator=g) * 31.0).clamp_max(30.0)
    ew = -torch.floor(torch.rand(N, 1, device=device, generator=g) * 31.0).clamp_max(30.0)
    a2, w2 = a * torch.pow(2.0, ea), w * torch.pow(2.0, ew)
    ref2 = a2.double() @ w2.double().t()
    mag2 = (a2.double().abs() @ w2.double().abs().t()).clamp_min(1e-300)
    rows = {"what": "rows of both operands scaled by 2^-U{0..30}; max over all elements of |c - fp64| / sum_k |a||w| of that element"}
    try:
        for mode in ("x3", "x2", "fp32"):
            _lib.set_gemm_mode(mode)
            check(L.pdgn_gemm_nt(ctypes.c_longlong(M), N, K, ptr(a), K, ptr(w), K, None, None, 0, ptr(c), N, None, stream_of(a)),
                  "pdgn_gemm_nt")
            out["max_error_vs_fp64_" + mode] = ((c.double() - ref).abs() / mag).max().item()
            check(L.pdgn_gemm_nt(ctypes.c_longlong(M), N, K, ptr(a2), K, ptr(w2), K, None, None, 0, ptr(c), N, None, stream_of(a)),
                  "pdgn_gemm_nt")
            rows["max_error_vs_fp64_" + mode] = ((c.double() - ref2).abs() / mag2).max().item()
    finally:
        _lib.set_gemm_mode(saved)
    out["row_scaled"] = rows
    return out


def _nt_entry(label, M, N, K, device):
    a = torch.randn(M, K, device=device)
    w = torch.randn(N, K, device=device)
    c = torch.empty(M, N, device=device)
    L = _lib.lib()
    _lib.ensure_scale_slots(torch.device(device))
    x3 = gemm_mode() != "fp32"
    # as the step launches it: on the bf16 matrix cores the block's assembled weight arrives pre-split (fused.split_planes, once
    # per iteration) and the kernel is the PW instance of gemm_x3_kernel (pdgn_gemm_nt_ps)
    from .fused import split_planes
    planes = split_planes(w, False, rows=M) if x3 else None      # (rows: two fp16 parts where the mode in force runs this shape on them)

    from .fused import _tail_workspace

    def run():
        ws = _tail_workspace(L, M, N, K, False, device, parts=planes.p.shape[0] if planes is not None else None)      # the stream-K tail without atomics, as the step's calls run it
        if planes is not None:
            P = planes.p
            check(L.pdgn_gemm_nt_ps(ctypes.c_longlong(M), N, K, ptr(a), K, ptr(P), P.shape[2], ctypes.c_longlong(P.shape[1] * P.shape[2]),
                                    P.shape[0], None, None, 0, ptr(c), N, None, None, 0, 1, 0, None, 0, stream_of(a)), "pdgn_gemm_nt_ps")
        else:
            check(L.pdgn_gemm_nt(ctypes.c_longlong(M), N, K, ptr(a), K, ptr(w), K, None, None, 0, ptr(c), N, None, stream_of(a)),
                  "pdgn_gemm_nt")
    us = _time_us(run)
    e = _entry("%s (%s, M=%d N=%d K=%d%s)" % ("gemm_x3_kernel" if x3 else "gemm_nt_kernel", label, M, N, K,
                                               "; PW instance = pdgn_gemm_nt_ps, the weight pre-split once per iteration" if planes is not None else ""),
               "mfma", 2.0 * M * N * K, us, x3=x3, two=planes is not None and planes.parts_p == 2, shape=[M, N, K])
    e["algorithmic_bytes_per_launch"] = 4.0 * (M * N + M * K + N * K)
    return e


def conv2_dense_stage4(B, base_points, device):
    """The largest single contraction of the step: conv2's dense half at stage 4 (models/PDGNet_v2.py:644 after the
    re-association of DESIGN.md section 3), out (M x 512) = (inte*w) (M x 5120) Wb^T over M = B * 8*base rows, on
    pdgn_gemm_nt.  Algorithmic flops 2*M*N*K; bytes (M*N + M*K + N*K) * 4."""
    return _nt_entry("conv2 dense half forward, stage 4", B * 8 * base_points, 512, 5120, device)


def per_point_stage4(B, base_points, device):
    """The per-point GEMM of stage 4: Y (M x 12832) = X (M x 128) Wcat^T, all taps of inte_conv_hk / conv2 / conv_fea.  63 flop per
    byte: under the two-part form's matrix roof (833 TFLOP/s) this launch is bound by its 1.84 GB of result STORES, not by its
    118 GFLOP (it was priced as `mfma`) -- bound "hbm", algorithmic bytes = result + both operands.  Round 6: on
    the row-panel kernel (csrc/gemm_rp.hip) in the default mode."""
    M, N, K = B * 8 * base_points, 12832, 128
    e = _nt_entry("per-point GEMM, stage 4", M, N, K, device)
    rp = gemm_mode() == "x2"
    h = _entry(e["kernel"].replace("gemm_x3_kernel", "gemm_rp_kernel<128>") if rp else e["kernel"], "hbm", 4.0 * (M * N + M * K + N * K),
               e["us_per_launch"], shape=[M, N, K])
    h["mfma_view"] = {"achieved_tflops": e["achieved"], "frac_of_matrix_roof": e["frac"], "peak_tflops": e["peak"]}
    return h


def conv2_dense_dx_stage4(B, base_points, device):
    """Input gradient of conv2's dense half: d(inte*w) (M x 5120) = dout (M x 512) Wb (512 x 5120) on pdgn_gemm_nn -- the
    kernel instance the step launches for it (the layer's weight as the transposed operand)."""
    M, N, K = B * 8 * base_points, 5120, 512
    dy = torch.randn(M, K, device=device)
    wb = torch.randn(K, N, device=device)                   # the layer's own (512 x 5120) weight: the transposed operand
    dx = torch.empty(M, N, device=device)
    L = _lib.lib()

    def run():
        check(L.pdgn_gemm_nn(ctypes.c_longlong(M), N, K, ptr(dy), K, ptr(wb), N, None, None, 0, ptr(dx), N, None, stream_of(dy)),
              "pdgn_gemm_nn")
    us = _time_us(run)
    x3 = gemm_mode() != "fp32"
    e = _entry("%s<WT> = pdgn_gemm_nn (conv2 dense half input gradient, stage 4, M=%d N=%d K=%d)"
               % ("gemm_x3_kernel" if x3 else "gemm_nt_kernel", M, N, K), "mfma", 2.0 * M * N * K, us, x3=x3,
               two=_two_part(M, N, K, (M * K + N * K) * 4), shape=[M, N, K])
    e["algorithmic_bytes_per_launch"] = 4.0 * (M * N + M * K + N * K)
    return e


def weight_grad_stage4(B, base_points, device):
    """conv2's dense half weight gradient at stage 4, dW (512 x 5120) = dY^T (inte*w) over M = B * 8*base rows, on the entry
    point the step launches for it: pdgn_gemm_tn_big (the x3 kernel with both operands transposed, stream-K over the rows;
    the launch zero-fills dW itself) -- or, with PDGN_GEMM=fp32, pdgn_gemm_tn (split row reduction, fp32 instructions)."""
    M, N, K = B * 8 * base_points, 512, 5120
    dy = torch.randn(M, N, device=device)
    x = torch.randn(M, K, device=device)
    dw = torch.zeros(N, K, device=device)
    L = _lib.lib()
    if gemm_mode() != "fp32":
        def run():
            check(L.pdgn_gemm_tn_big(ctypes.c_longlong(M), N, K, ptr(dy), N, ptr(x), K, ptr(dw), 0, stream_of(dy)), "pdgn_gemm_tn_big")
        us = _time_us(run)
        e = _entry("gemm_x3_kernel<AT,WT> = pdgn_gemm_tn_big (dW of conv2's dense half, stage 4, M=%d N=%d K=%d)" % (M, N, K), "mfma",
                   2.0 * M * N * K, us, x3=True, two=_two_part(N, K, M, (M * N + M * K) * 4), shape=[M, N, K])
        e["algorithmic_bytes_per_launch"] = 4.0 * (M * N + M * K + N * K)
        return e

    def run():
        dw.zero_()
        check(L.pdgn_gemm_tn(ctypes.c_longlong(M), N, K, ptr(dy), ptr(x), ptr(dw), stream_of(dy)), "pdgn_gemm_tn")
    # the zero-fill of dW is part of the price (pdgn_gemm_tn accumulates split partial sums with atomics): it is timed
    # WITH the kernel; `kernel_only_us` (the fill timed alone subtracted) is what rocprofv3's kernel trace reports
    us = _time_us(run)
    fill_us = _time_us(lambda: dw.zero_())
    e = _entry("gemm_tn_kernel (dW of conv2's dense half, stage 4, M=%d N=%d K=%d)" % (M, N, K), "mfma", 2.0 * M * N * K, us,
               shape=[M, N, K], kernel_only_us=us - fill_us)
    e["algorithmic_bytes_per_launch"] = 4.0 * (M * N + M * K + N * K)
    return e


def bn_act_backward_stage4(B, base_points, device):
    """BatchNorm+LeakyReLU(+product) backward of inte_conv_hk's output at stage 4: rows = B*N*5,
    C = 1024.  Algorithmic bytes: reduce reads x, dy, mul; apply reads x, dy, mul and writes dx, dmul
    = 8 * rows * C * 4."""
    rows, C = B * 8 * base_points * 5, 1024
    x = torch.randn(rows, C, device=device)
    dy = torch.randn(rows, C, device=device)
    mul = torch.rand(rows, C, device=device)
    stats = torch.cat([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).to(device)
    L = _lib.lib()
    L.pdgn_bn_scratch_floats.restype = ctypes.c_longlong
    scratch = torch.empty(L.pdgn_bn_scratch_floats(ctypes.c_longlong(rows), C), device=device)
    bs = torch.empty(2 * C, device=device)
    dx, dmul = torch.empty_like(x), torch.empty_like(x)

    def run():
        check(L.pdgn_bn_act_backward(ctypes.c_longlong(rows), C, 2, 1, ptr(x), ptr(dy), ptr(mul), ptr(stats),
                                     ptr(scratch), ptr(bs), ptr(dx), ptr(dmul), 0, stream_of(x)), "pdgn_bn_act_backward")
    us = _time_us(run)
    return _entry("cl_bwd_reduce + cl_bwd_apply (BN+LeakyReLU*w backward, rows=%d C=%d)" % (rows, C), "hbm",
                  8.0 * rows * C * 4, us)


def window_gather_sum_stage4(B, base_points, device):
    """inte_conv_hk's gather half at the last stage: N = 8*base points, F = 256, k = 10, T = 6 taps, P = 5 positions,
    C = 4F.  Algorithmic (compulsory) bytes per point: the T + 1 segments of its Y row read ONCE, P written rows of C
    floats, k indices.  The kernel issues T*P + 1 row loads per point (`gathered_bytes_per_launch`): each Y segment
    is wanted by ~P windows of other points; the task mapping of wgs_fwd_xcd_kernel serves those re-reads from one
    XCD's L2 (before it they were L2 misses: 5.4 GB of fabric traffic per launch, 704 us)."""
    N, F, k, T, P = 8 * base_points, 256, 10, 6, 5
    C = 4 * F
    ldy = T * C + C
    Y = torch.randn(B, N, ldy, device=device)
    idx = torch.randint(0, N, (B, N, k), device=device, dtype=torch.int32)
    out = torch.empty(B, N, P, C, device=device)
    L = _lib.lib()

    def run():
        check(L.pdgn_window_gather_sum(B, N, k, ldy, T, P, C, 0, T * C, ptr(Y), ptr(idx), None, 0, ptr(out),
                                       stream_of(Y)), "pdgn_window_gather_sum")
    us = _time_us(run)
    return _entry("wgs_fwd_xcd_kernel<6, 32> (inte_conv_hk gather, stage 4)", "hbm",
                  float(B * N * ((T + 1 + P) * C * 4 + k * 4)), us,
                  gathered_bytes_per_launch=float(B * N * ((T * P + 1 + P) * C * 4 + k * 4)))


def feature_knn_stage4(B, base_points, device):
    """Feature-space kNN of the last stage: Gram flops = 2*N*N*F per sample on the fp32 MFMA.  F = the 128 channels that VARY over a
    sample's points: the 128 broadcast ones cancel in every pairwise distance and stay out of the graph (deconv.start_feature_knn)."""
    N, F, k = 8 * base_points, 128, 10
    x = torch.randn(B, F, N, device=device)
    idx = torch.empty(B, N, k, device=device, dtype=torch.int32)
    sq = torch.empty(B, N, device=device)
    L = _lib.lib()

    def run():
        check(L.pdgn_feature_knn(B, F, N, k, ptr(x), ptr(sq), ptr(idx), stream_of(x)), "pdgn_feature_knn")
    us = _time_us(run)
    return _entry("feat_knn_pc_kernel<64> (stage-4 kNN graph, 128 varying channels)", "mfma", 2.0 * B * N * N * F, us)


def knn3_largest(B, base_points, device):
    """3-D kNN of the local-pair loss at its largest call (n = 16*base, m = 8*base, k = 20):
    algorithmic bytes B*(12n + 12m + 8mk); VALU-bound."""
    n, m, k = 16 * base_points, 8 * base_points, 20
    xyz = torch.randn(B, n, 3, device=device)
    q = xyz[:, :m].contiguous()
    idx = torch.empty(B, m, k, device=device, dtype=torch.int32)
    d2 = torch.empty(B, m, k, device=device)
    L = _lib.lib()

    def run():
        check(L.pdgn_knnquery(B, n, m, k, ptr(xyz), ptr(q), ptr(idx), ptr(d2), stream_of(xyz)), "pdgn_knnquery")
    us = _time_us(run)
    e = _entry("knn3_wave4_kernel (n=%d, m=%d, k=20)" % (n, m), "hbm", float(B * (12 * n + 12 * m + 8 * m * k)), us)
    e["distance_evals_per_s"] = B * n * m / us * 1e6
    e["note"] = ("selection-bound, not bandwidth-bound (intensity ~80 flop/B): vector-ALU issue utilisation and wait "
                 "fractions are in the round's profiles/r*_pmc_summary.txt (knn3_largest row); the HBM fraction is reported because the north star asks for it")
    return e


def emd_cost_c5(B, base_points, device, pairs=512):
    """Config C5's dominant kernel (not part of a training step: the target of the eval PMC pass, tools/run_pmc_roofline.sh):
    the fused approximate-EMD cost on `pairs` pairs of 2048 x 2048 points."""
    from .structural_losses import emd_cost
    g = torch.Generator().manual_seed(9999)
    a = (torch.rand(pairs, 2048, 3, generator=g) * 2 - 1).to(device)
    b = (torch.rand(pairs, 2048, 3, generator=g) * 2 - 1).to(device)
    us = _time_us(lambda: emd_cost(a, b))
    return {"kernel": "emd_cost_kernel (%d pairs of 2048 x 2048)" % pairs, "bound": "valu-issue", "us_per_launch": us,
            "pairs_per_s": pairs / us * 1e6}


def conv2_in_step_spans(launch_list, B, base_points):
    """Spans of a recorded iteration (PDGNTrainer.capture_list) that ARE the dominant contraction: conv2's dense half at stage 4,
    forward, one per generator pass.  The data-parallel launch is found by the kernel INSTANCE and grid pdgn_gemm_nt_ps uses for
    that problem (pdgn_gemm_nt_ps_launch_info); the span takes in what belongs to the same call on the same stream: the memset in
    front of it and the stream-K tail launch behind it (560 tiles on 256 CUs: two whole rounds + a tail of 48 tiles).  [] when
    the instance cannot be named (fp32 mode, the 16x16x32 arm)."""
    if gemm_mode() == "fp32":
        return []
    from .fused import two_part
    L = _lib.lib()
    M = B * 8 * base_points
    from .fused import two_part_planes
    parts = 2 if two_part_planes(M, 512, 5120, 0) else 3          # (the step's planes: inte arrives with its maxima)
    sym, grid, red, scan = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_void_p(), ctypes.c_void_p()
    if L.pdgn_gemm_nt_ps_launch_info(ctypes.c_longlong(M), 512, 5120, parts, ctypes.byref(sym), ctypes.byref(grid)) != 0 or not sym.value:
        return []
    L.pdgn_gemm_aux_symbols(ctypes.byref(red), ctypes.byref(scan))
    cfg = L.pdgn_gemm_nt_config(ctypes.c_longlong(M), 512, 5120, 0)
    spans = []
    for pos in launch_list.kernel_nodes(sym.value, grid.value):
        first = last = pos
        p, kind, ksym = launch_list.neighbor(pos, -1)             # in front: the zero-fill of an atomic tail, or the scan of the activations (two parts)
        if p >= 0 and ((cfg >= 16 and kind == 1) or (kind == 0 and ksym == scan.value)):
            first = p
        if cfg >= 16:                                            # a stream-K tail follows: data-parallel | tail | reduce of its partial tiles
            p, kind, _ = launch_list.neighbor(pos, +1)
            if p >= 0 and kind == 0:
                last = p
                p2, kind2, ksym2 = launch_list.neighbor(p, +1)
                if p2 >= 0 and kind2 == 0 and ksym2 == red.value:
                    last = p2
        spans.append((first, last))
    return spans


def attach_in_step(top, launches_ms):
    """The dominant kernel's roofline entry from its launches INSIDE the timed steps (HIP events on the stream it is launched on,
    around every launch: csrc/replay.hip); the stand-alone figure of `measure` (20 launches back to back) moves to `back_to_back`."""
    ms = [v for node in launches_ms for v in node]
    if not ms:
        return top
    us = sum(ms) / len(ms) * 1e3
    work = top["algorithmic_flops_per_launch"]
    out = dict(top)
    out["back_to_back"] = {"us_per_launch": top["us_per_launch"], "achieved": top["achieved"], "frac": top["frac"],
                           "what": "the same call alone, 20 back to back after 3 (the chip's power limit holds ~1.7 GHz there, between "
                                   "the iteration's bandwidth-bound kernels it clocks higher; a two-part call scans its first operand's "
                                   "maxima here, ~150 us of the figure, which the step gets from the producing kernel)"}
    out["us_per_launch"] = us
    out["achieved"] = work / us / 1e6
    out["frac"] = out["achieved"] / out["peak"]
    out["timing"] = ("HIP events around each of the contraction's %d calls inside the timed steps (one per generator pass and step: "
                     "data-parallel launch + stream-K tail launch + the reduce of its partial tiles, as pdgn_gemm_nt_ps issues them; a scan "
                     "of the activations' maxima in front where the call makes one -- in the step the kernel that writes them leaves "
                     "the maxima behind), on the stream the launch list issues them on; min %.1f / max %.1f us"
                     % (len(ms), min(ms) * 1e3, max(ms) * 1e3))
    if "mfma" in out and "executed_tflops" in out["mfma"]:
        out["mfma"] = dict(out["mfma"], executed_tflops=out["achieved"] * out["mfma"].get("products_per_fp32_product", X3_PRODUCTS))
    return out


ENTRIES = (conv2_dense_stage4, per_point_stage4, conv2_dense_dx_stage4, weight_grad_stage4, bn_act_backward_stage4,
           window_gather_sum_stage4, feature_knn_stage4, knn3_largest)


def measure(B, base_points, device):
    """Roofline object of the dominant kernel of the step -- its largest single contraction, conv2's dense half at
    stage 4 on pdgn_gemm_nt -- with the other hand-written kernels under "others"."""
    entries = [f(B, base_points, device) for f in ENTRIES]
    # `traffic` is NOT measured by this run: it is the PMC figure (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes,
    # corrected as the guide prescribes) committed in profiles/traffic.json by tools/pmc_roofline.py, attached only when
    # that file was recorded at this batch / resolution
    try:
        with open(_TRAFFIC) as f:
            traffic = json.load(f)
        meta = traffic.get("_meta", {})
        if meta.get("batch", 35) == B and meta.get("base_points", 128) == base_points:
            for e in entries:
                for key, val in traffic.items():
                    if key != "_meta" and e["kernel"].startswith(key):
                        e["traffic"] = val
                        e["traffic_source"] = "profiles/traffic.json (%s)" % meta.get("recorded", "round 1 PMC passes")
    except (OSError, ValueError):
        pass
    top = dict(entries[0])
    top["others"] = entries[1:]
    return top
