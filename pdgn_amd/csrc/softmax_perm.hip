// softmax_perm.hip -- softmax over the k neighbour slots fused with the reference's slot/channel
// interleave (models/PDGNet_v2.py:634-641): h (M, k, C) point-major bilateral logits (after
// conv_all's BatchNorm+LeakyReLU) ->
//     w_r[m, p, 2c + j] = softmax_s(h[m, :, c])[s = P*j + p],   P = k/2,
// i.e. directly in the (M, P, 2C) layout in which inte_conv_hk's output is multiplied (:642).
// One thread owns one (m, c): k strided reads (coalesced across c), P float2 writes.  The adjoint
// reads w_r / dw_r the same way:  dh_s = w_s (dw_s - sum_s' w_s' dw_s').
#include "common.h"

#define SP_THREADS 256
#define SP_MAXK 32

__global__ __launch_bounds__(SP_THREADS) void softmax_perm_fwd_kernel(long long total, int k, int C,
                                                                      const float *__restrict__ h,
                                                                      float *__restrict__ w) {
    const long long e = (long long)blockIdx.x * SP_THREADS + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C);
    const long long m = e / C;
    const int P = k / 2;
    const float *H = h + m * k * C + c;
    float v[SP_MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < SP_MAXK; ++s)
        if (s < k) { v[s] = H[(size_t)s * C]; mx = fmaxf(mx, v[s]); }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < SP_MAXK; ++s)
        if (s < k) { v[s] = __expf(v[s] - mx); sum += v[s]; }
    const float inv = 1.0f / sum;
    float *W = w + m * k * C + 2 * c;                       // row (m, p): 2C floats
#pragma unroll
    for (int p = 0; p < SP_MAXK / 2; ++p)
        if (p < P) *reinterpret_cast<float2 *>(W + (size_t)p * 2 * C) = make_float2(v[p] * inv, v[P + p] * inv);
}

__global__ __launch_bounds__(SP_THREADS) void softmax_perm_bwd_kernel(long long total, int k, int C,
                                                                      const float *__restrict__ w,
                                                                      const float *__restrict__ dw,
                                                                      float *__restrict__ dh) {
    const long long e = (long long)blockIdx.x * SP_THREADS + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C);
    const long long m = e / C;
    const int P = k / 2;
    const float *W = w + m * k * C + 2 * c, *G = dw + m * k * C + 2 * c;
    float wv[SP_MAXK], gv[SP_MAXK];
    float dot = 0.f;
#pragma unroll
    for (int p = 0; p < SP_MAXK / 2; ++p)
        if (p < P) {
            const float2 a = *reinterpret_cast<const float2 *>(W + (size_t)p * 2 * C);
            const float2 g = *reinterpret_cast<const float2 *>(G + (size_t)p * 2 * C);
            wv[p] = a.x; wv[P + p] = a.y; gv[p] = g.x; gv[P + p] = g.y;
            dot = __fmaf_rn(a.x, g.x, __fmaf_rn(a.y, g.y, dot));
        }
    float *D = dh + m * k * C + c;
#pragma unroll
    for (int s = 0; s < SP_MAXK; ++s)
        if (s < k) D[(size_t)s * C] = wv[s] * (gv[s] - dot);
}

extern "C" int pdgn_softmax_slots_permute(long long m, int k, int c, const float *h, float *w, pdgn_stream_t stream) {
    if (m < 1 || k < 2 || k > SP_MAXK || (k & 1) || c < 1) return PDGN_ERR_INVALID;
    const long long total = m * c;
    hipLaunchKernelGGL(softmax_perm_fwd_kernel, dim3(cdiv(total, SP_THREADS)), dim3(SP_THREADS), 0, (hipStream_t)stream,
                       total, k, c, h, w);
    return pdgn_launch_status();
}

extern "C" int pdgn_softmax_slots_permute_backward(long long m, int k, int c, const float *w, const float *dw,
                                                   float *dh, pdgn_stream_t stream) {
    if (m < 1 || k < 2 || k > SP_MAXK || (k & 1) || c < 1) return PDGN_ERR_INVALID;
    const long long total = m * c;
    hipLaunchKernelGGL(softmax_perm_bwd_kernel, dim3(cdiv(total, SP_THREADS)), dim3(SP_THREADS), 0, (hipStream_t)stream,
                       total, k, c, w, dw, dh);
    return pdgn_launch_status();
}
