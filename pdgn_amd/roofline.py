"""Live roofline measurement of the hand-written kernels at the launch shapes of one G+D step.

Each entry launches ONE kernel of libpdgn_hip.so through the C ABI on torch's current stream
and times it with HIP events recorded on that same stream (torch.cuda.Event == hipEvent on
ROCm).  `achieved` = algorithmic bytes (or flops) per launch / average launch duration; the
per-unit figures are stated in DESIGN.md ("Kernels and rooflines").
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_of

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA peak


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


def window_gather_sum_stage4(B, base_points, device):
    """inte_conv_hk's gather half at the last stage: N = 8*base points, F = 256, k = 10,
    T = 6 taps, P = 5 positions, C = 4F.  Algorithmic bytes per point: 30 gathered rows + 1 centre
    row + 5 written rows of C floats, + 10 indices."""
    N, F, k, T, P = 8 * base_points, 256, 10, 6, 5
    C = 4 * F
    ldy = T * C + C
    Y = torch.randn(B, N, ldy, device=device)
    idx = torch.randint(0, N, (B, N, k), device=device, dtype=torch.int32)
    out = torch.empty(B, N, P, C, device=device)
    L = _lib.lib()

    def run():
        check(L.pdgn_window_gather_sum(B, N, k, ldy, T, P, C, 0, T * C, ptr(Y), ptr(idx), None, ptr(out),
                                       stream_of(Y)), "pdgn_window_gather_sum")
    us = _time_us(run)
    bytes_ = B * N * ((T * P + 1 + P) * C * 4 + k * 4)
    return {"kernel": "wgs_fwd_kernel<4> (inte_conv_hk gather, stage 4)", "bound": "hbm",
            "achieved": bytes_ / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_ / us / 1e3 / HBM_PEAK_GBS, "traffic": None, "us_per_launch": us,
            "algorithmic_bytes_per_launch": bytes_}


def feature_knn_stage4(B, base_points, device):
    """Feature-space kNN of the last stage: Gram flops = 2*N*N*F per sample on the fp32 MFMA."""
    N, F, k = 8 * base_points, 256, 10
    x = torch.randn(B, F, N, device=device)
    idx = torch.empty(B, N, k, device=device, dtype=torch.int32)
    sq = torch.empty(B, N, device=device)
    L = _lib.lib()

    def run():
        check(L.pdgn_feature_knn(B, F, N, k, ptr(x), ptr(sq), ptr(idx), stream_of(x)), "pdgn_feature_knn")
    us = _time_us(run)
    flops = 2.0 * B * N * N * F
    return {"kernel": "feat_knn_kernel<128> (stage-4 kNN graph)", "bound": "mfma",
            "achieved": flops / us / 1e6, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flops / us / 1e6 / MFMA_F32_PEAK_TFLOPS, "traffic": None, "us_per_launch": us,
            "algorithmic_flops_per_launch": flops}


def measure(B, base_points, device):
    """Roofline object of the dominant hand-written kernel (+ the runners-up under "others")."""
    entries = [window_gather_sum_stage4(B, base_points, device), feature_knn_stage4(B, base_points, device)]
    entries.sort(key=lambda e: -e["us_per_launch"])
    top = dict(entries[0])
    top["others"] = entries[1:]
    return top
