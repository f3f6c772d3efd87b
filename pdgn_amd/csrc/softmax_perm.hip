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

// Same, with the BatchNorm + activation that precedes the softmax folded in (conv_all.4 + LeakyReLU,
// models/PDGNet_v2.py:623-625): x is the RAW conv output, stats = [scale | shift | ...] of pdgn_bn_stats; the
// activated logits never reach HBM.  One thread owns a channel PAIR: float2 reads, float4 writes.
__device__ __forceinline__ float sp_act(float z, int act) {
    return act == 2 ? (z > 0.f ? z : 0.01f * z) : (act == 1 ? fmaxf(z, 0.f) : z);
}

template <int KT>   // compile-time k (0 = runtime k <= SP_MAXK): branch-free unrolled loads, all k in flight at once
__global__ __launch_bounds__(SP_THREADS) void bn_softmax_perm_fwd_kernel(long long total2, int k_rt, int C, int act,
                                                                         const float *__restrict__ x,
                                                                         const float *__restrict__ stats,
                                                                         float *__restrict__ w) {
    const long long e = (long long)blockIdx.x * SP_THREADS + threadIdx.x;
    if (e >= total2) return;
    constexpr int KM = KT ? KT : SP_MAXK;
    const int k = KT ? KT : k_rt;
    const int C2 = C / 2, c = (int)(e % C2) * 2;
    const long long m = e / C2;
    const int P = k / 2;
    const float2 sc = *reinterpret_cast<const float2 *>(stats + c), sh = *reinterpret_cast<const float2 *>(stats + C + c);
    const float *H = x + m * k * C + c;
    float2 v[KM];
    float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
    for (int s = 0; s < KM; ++s)
        if (KT || s < k) v[s] = *reinterpret_cast<const float2 *>(H + (size_t)s * C);
#pragma unroll
    for (int s = 0; s < KM; ++s)
        if (KT || s < k) {
            const float2 r = v[s];
            v[s].x = sp_act(__fmaf_rn(r.x, sc.x, sh.x), act);
            v[s].y = sp_act(__fmaf_rn(r.y, sc.y, sh.y), act);
            mx0 = fmaxf(mx0, v[s].x);
            mx1 = fmaxf(mx1, v[s].y);
        }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int s = 0; s < KM; ++s)
        if (KT || s < k) {
            v[s].x = __expf(v[s].x - mx0);
            v[s].y = __expf(v[s].y - mx1);
            s0 += v[s].x;
            s1 += v[s].y;
        }
    const float i0 = 1.0f / s0, i1 = 1.0f / s1;
    float *W = w + m * k * C + 2 * c;                       // row (m, p): 2C floats; [2c, 2c+1, 2c+2, 2c+3]
#pragma unroll
    for (int p = 0; p < KM / 2; ++p)
        if (KT || p < P)
            *reinterpret_cast<float4 *>(W + (size_t)p * 2 * C) =
                make_float4(v[p].x * i0, v[P + p].x * i0, v[p].y * i1, v[P + p].y * i1);
}

// The bilateral weighting of a deconvolution block in ONE pass (models/PDGNet_v2.py:623-642):
//   w = softmax_slots(act(BN_a(x)))  (interleaved layout),   y = act_i(BN_i(u)) * w
// x (M, k, C) raw conv_all.3 output, u (M, k/2, 2C) raw inte_conv_hk output -- already in w's layout, so the thread
// that owns channel pair c of point m holds exactly the float4s of u it has to scale.  w is written only when the
// backward pass will need it (w_out != NULL); y always.  Neither activated tensor makes an extra HBM round trip.
// max_out (may be NULL): uint32[M], the maximum of |y| (bit pattern) over each point's k C outputs -- the ROW maxima of y as the
// (M, k C) first operand of conv2's dense half (combined with atomic max: order-independent; zero-filled by the launcher) -- what
// the two-part contraction that consumes y (gemm_x3.hip: one power-of-two scale per row) would otherwise scan y for.
template <int KT>
__global__ __launch_bounds__(SP_THREADS) void bn_softmax_perm_mul_fwd_kernel(
    long long total2, int k_rt, int C, int act, const float *__restrict__ x, const float *__restrict__ stats,
    int act_u, const float *__restrict__ u, const float *__restrict__ stats_u, float *__restrict__ w_out,
    float *__restrict__ y, unsigned *__restrict__ max_out, const float *__restrict__ gamma_u, const float *__restrict__ beta_u,
    float bound_scale, unsigned *__restrict__ cmax_out) {
    const long long e = (long long)blockIdx.x * SP_THREADS + threadIdx.x;
    unsigned ymax = 0u;
    if (e < total2) {
    constexpr int KM = KT ? KT : SP_MAXK;
    const int k = KT ? KT : k_rt;
    const int C2 = C / 2, c = (int)(e % C2) * 2;
    const long long m = e / C2;
    const int P = k / 2;
    const float2 sc = *reinterpret_cast<const float2 *>(stats + c), sh = *reinterpret_cast<const float2 *>(stats + C + c);
    const float4 su = *reinterpret_cast<const float4 *>(stats_u + 2 * c);
    const float4 hu = *reinterpret_cast<const float4 *>(stats_u + 2 * C + 2 * c);
    const float *H = x + m * k * C + c;
    const size_t o = (size_t)m * k * C + 2 * c;                // row (m, p): 2C floats; [2c .. 2c+3]
    if (cmax_out && m == 0) {
        // an upper bound of the COLUMN maxima of y as the (M, k C) operand (its weight gradient's scales): softmax weights <= 1 and,
        // under batch statistics over n samples, |xhat| <= sqrt(n - 1) = bound_scale, so |y[:, (p, j)]| <= |beta_j| + |gamma_j| bound_scale
        // -- from the 2C parameters alone, written by the threads of point 0 (the same for every slot p)
        const float4 g = *reinterpret_cast<const float4 *>(gamma_u + 2 * c), b = *reinterpret_cast<const float4 *>(beta_u + 2 * c);
        uint4 bd;
        bd.x = __float_as_uint(__fmaf_rn(fabsf(g.x), bound_scale, fabsf(b.x)));
        bd.y = __float_as_uint(__fmaf_rn(fabsf(g.y), bound_scale, fabsf(b.y)));
        bd.z = __float_as_uint(__fmaf_rn(fabsf(g.z), bound_scale, fabsf(b.z)));
        bd.w = __float_as_uint(__fmaf_rn(fabsf(g.w), bound_scale, fabsf(b.w)));
        for (int p = 0; p < P; ++p) *reinterpret_cast<uint4 *>(cmax_out + (size_t)p * 2 * C + 2 * c) = bd;
    }
    float2 v[KM];
    float4 uu[KM / 2];
#pragma unroll
    for (int s = 0; s < KM; ++s)
        if (KT || s < k) v[s] = *reinterpret_cast<const float2 *>(H + (size_t)s * C);
#pragma unroll
    for (int p = 0; p < KM / 2; ++p)
        if (KT || p < P) uu[p] = *reinterpret_cast<const float4 *>(u + o + (size_t)p * 2 * C);
    float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
    for (int s = 0; s < KM; ++s)
        if (KT || s < k) {
            const float2 r = v[s];
            v[s].x = sp_act(__fmaf_rn(r.x, sc.x, sh.x), act);
            v[s].y = sp_act(__fmaf_rn(r.y, sc.y, sh.y), act);
            mx0 = fmaxf(mx0, v[s].x);
            mx1 = fmaxf(mx1, v[s].y);
        }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int s = 0; s < KM; ++s)
        if (KT || s < k) {
            v[s].x = __expf(v[s].x - mx0);
            v[s].y = __expf(v[s].y - mx1);
            s0 += v[s].x;
            s1 += v[s].y;
        }
    const float i0 = 1.0f / s0, i1 = 1.0f / s1;
#pragma unroll
    for (int p = 0; p < KM / 2; ++p)
        if (KT || p < P) {
            const float4 wv = make_float4(v[p].x * i0, v[P + p].x * i0, v[p].y * i1, v[P + p].y * i1);
            if (w_out) *reinterpret_cast<float4 *>(w_out + o + (size_t)p * 2 * C) = wv;
            float4 r;
            r.x = sp_act(__fmaf_rn(uu[p].x, su.x, hu.x), act_u) * wv.x;
            r.y = sp_act(__fmaf_rn(uu[p].y, su.y, hu.y), act_u) * wv.y;
            r.z = sp_act(__fmaf_rn(uu[p].z, su.z, hu.z), act_u) * wv.z;
            r.w = sp_act(__fmaf_rn(uu[p].w, su.w, hu.w), act_u) * wv.w;
            *reinterpret_cast<float4 *>(y + o + (size_t)p * 2 * C) = r;
            ymax = max(max(ymax, max(__float_as_uint(r.x) & 0x7fffffffu, __float_as_uint(r.y) & 0x7fffffffu)),
                       max(__float_as_uint(r.z) & 0x7fffffffu, __float_as_uint(r.w) & 0x7fffffffu));
        }
    }
    if (max_out) {                                                 // (uniform: every thread of the workgroup gets here)
        // y viewed as the (M, k C) operand of conv2's dense half: the maximum of |y| over ROW m (all slots and channels of point m)
        const int C2 = C / 2;
        if ((C2 & 63) == 0) {                                      // a wave lies inside one point: one atomic per wave
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) ymax = max(ymax, (unsigned)__shfl_xor((int)ymax, o));
            if ((threadIdx.x & 63) == 0 && e < total2) atomicMax(max_out + e / C2, ymax);
        } else if (e < total2) {
            atomicMax(max_out + e / C2, ymax);
        }
    }
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

extern "C" int pdgn_bn_softmax_slots_permute(long long m, int k, int c, int act, const float *x, const float *stats,
                                             float *w, pdgn_stream_t stream) {
    if (m < 1 || k < 2 || k > SP_MAXK || (k & 1) || c < 2 || (c & 1) || act < 0 || act > 2) return PDGN_ERR_INVALID;
    const long long total2 = m * (c / 2);
    const dim3 grid(cdiv(total2, SP_THREADS)), block(SP_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (k == 10) hipLaunchKernelGGL(bn_softmax_perm_fwd_kernel<10>, grid, block, 0, s, total2, k, c, act, x, stats, w);
    else if (k == 20) hipLaunchKernelGGL(bn_softmax_perm_fwd_kernel<20>, grid, block, 0, s, total2, k, c, act, x, stats, w);
    else if (k == 4) hipLaunchKernelGGL(bn_softmax_perm_fwd_kernel<4>, grid, block, 0, s, total2, k, c, act, x, stats, w);
    else hipLaunchKernelGGL(bn_softmax_perm_fwd_kernel<0>, grid, block, 0, s, total2, k, c, act, x, stats, w);
    return pdgn_launch_status();
}

extern "C" int pdgn_bn_softmax_slots_permute_mul(long long m, int k, int c, int act, const float *x, const float *stats,
                                                 int act_u, const float *u, const float *stats_u, float *w, float *y,
                                                 unsigned *max_out, const float *gamma_u, const float *beta_u, float bound_scale,
                                                 unsigned *cmax_out, pdgn_stream_t stream) {
    if (m < 1 || k < 2 || k > SP_MAXK || (k & 1) || c < 2 || (c & 1) || act < 0 || act > 2 || act_u < 0 || act_u > 2)
        return PDGN_ERR_INVALID;
    if (cmax_out && (!gamma_u || !beta_u || !(bound_scale >= 0.f))) return PDGN_ERR_INVALID;
    const long long total2 = m * (c / 2);
    const dim3 grid(cdiv(total2, SP_THREADS)), block(SP_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (max_out && hipMemsetAsync(max_out, 0, (size_t)m * sizeof(unsigned), s) != hipSuccess) return pdgn_launch_status();
    if (k == 10)
        hipLaunchKernelGGL(bn_softmax_perm_mul_fwd_kernel<10>, grid, block, 0, s, total2, k, c, act, x, stats, act_u, u, stats_u, w, y, max_out, gamma_u, beta_u, bound_scale, cmax_out);
    else if (k == 20)
        hipLaunchKernelGGL(bn_softmax_perm_mul_fwd_kernel<20>, grid, block, 0, s, total2, k, c, act, x, stats, act_u, u, stats_u, w, y, max_out, gamma_u, beta_u, bound_scale, cmax_out);
    else if (k == 4)
        hipLaunchKernelGGL(bn_softmax_perm_mul_fwd_kernel<4>, grid, block, 0, s, total2, k, c, act, x, stats, act_u, u, stats_u, w, y, max_out, gamma_u, beta_u, bound_scale, cmax_out);
    else
        hipLaunchKernelGGL(bn_softmax_perm_mul_fwd_kernel<0>, grid, block, 0, s, total2, k, c, act, x, stats, act_u, u, stats_u, w, y, max_out, gamma_u, beta_u, bound_scale, cmax_out);
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
