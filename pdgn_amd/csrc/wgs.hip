// wgs.hip -- "window gather-sum": the gather half of the re-associated edge convolutions.
//
// The reference convolves materialised edge tensors e[b,c,n,s] = [x_n, x_idx(n,s) - x_n]
// (models/PDGNet_v2.py:462-477) with 1xT kernels sliding over the k neighbour slots
// (inte_conv_hk :562/:622, conv2 :559/:602, conv_fea :609).  Convolution is linear, so
//     sum_c sum_t W[o,c,t] e[c,n,p+t]  =  sum_t (W2_t X)[o, idx(n,p+t)]  +  ((sum_t W1_t - W2_t) X)[o,n]
// where Y = [W2_0 X | .. | W2_{T-1} X | Wc X] is ONE dense per-point GEMM (MFMA) and what remains is
//     out[b,n,p,c] = bias[c] + Y[b,n,offc+c] + sum_{t<T} Y[b, idx[b,n,p+t], off + t*C + c]
// -- this kernel.  It is a pure row gather (HBM/L2-bound): point-major rows, channels contiguous,
// float4 per lane, the neighbour index wave-uniform.
// Backward scatters dout rows back onto dY with contiguous 256-B wave atomics.
#include "common.h"

#define WGS_THREADS 256

template <int VEC>
__global__ __launch_bounds__(WGS_THREADS) void wgs_fwd_kernel(
    long long total, int n, int k, int ldy, int T, int P, int CV, int off, int offc,
    const float *__restrict__ Y, const int32_t *__restrict__ idx, const float *__restrict__ bias,
    float *__restrict__ out) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    long long e = (long long)blockIdx.x * WGS_THREADS + threadIdx.x;
    if (e >= total) return;
    const int cv = (int)(e % CV);
    long long r = e / CV;
    const int p = (int)(r % P);
    const long long bn = r / P;                    // b * n + point
    const long long b0 = bn / n * n;               // first point row of this batch
    const int c = cv * VEC;
    const int32_t *I = idx + bn * k + p;
    vec_t acc;
    if (bias) acc = *reinterpret_cast<const vec_t *>(bias + c);
    else for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (offc >= 0) acc += *reinterpret_cast<const vec_t *>(Y + bn * ldy + offc + c);
    for (int t = 0; t < T; ++t) {
        const long long row = b0 + I[t];
        acc += *reinterpret_cast<const vec_t *>(Y + row * ldy + off + t * CV * VEC + c);
    }
    *reinterpret_cast<vec_t *>(out + e * VEC) = acc;
}

__global__ __launch_bounds__(WGS_THREADS) void wgs_bwd_kernel(
    long long total, int n, int k, int ldy, int T, int P, int C, int off, int offc,
    const float *__restrict__ dout, const int32_t *__restrict__ idx, float *__restrict__ dY) {
    long long e = (long long)blockIdx.x * WGS_THREADS + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C);
    long long r = e / C;
    const int p = (int)(r % P);
    const long long bn = r / P;
    const long long b0 = bn / n * n;
    const float g = dout[e];
    const int32_t *I = idx + bn * k + p;
    for (int t = 0; t < T; ++t)
        atomicAdd(dY + (b0 + I[t]) * ldy + off + t * C + c, g);
    if (offc >= 0 && p == 0) {                      // centre columns have a single writer
        float s = g;
        for (int pp = 1; pp < P; ++pp) s += dout[e + (long long)pp * C];
        dY[bn * ldy + offc + c] = s;
    }
}

static bool wgs_ok(int b, int n, int k, int ldy, int T, int P, int C, int off, int offc) {
    return b >= 0 && n >= 1 && k >= 1 && T >= 1 && P >= 1 && C >= 1 && T + P - 1 <= k && off >= 0 &&
           off + T * C <= ldy && (offc < 0 || offc + C <= ldy);
}

extern "C" int pdgn_window_gather_sum(int b, int n, int k, int ldy, int T, int P, int C, int off, int offc,
                                      const float *Y, const int32_t *idx, const float *bias, float *out,
                                      pdgn_stream_t stream) {
    if (!wgs_ok(b, n, k, ldy, T, P, C, off, offc)) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = (C % 4 == 0) && (ldy % 4 == 0) && (off % 4 == 0) && (offc < 0 || offc % 4 == 0);
    if (v4) {
        long long total = (long long)b * n * P * (C / 4);
        hipLaunchKernelGGL(wgs_fwd_kernel<4>, dim3(cdiv(total, WGS_THREADS)), dim3(WGS_THREADS), 0, s, total,
                           n, k, ldy, T, P, C / 4, off, offc, Y, idx, bias, out);
    } else {
        long long total = (long long)b * n * P * C;
        hipLaunchKernelGGL(wgs_fwd_kernel<1>, dim3(cdiv(total, WGS_THREADS)), dim3(WGS_THREADS), 0, s, total,
                           n, k, ldy, T, P, C, off, offc, Y, idx, bias, out);
    }
    return pdgn_launch_status();
}

extern "C" int pdgn_window_gather_sum_backward(int b, int n, int k, int ldy, int T, int P, int C, int off,
                                               int offc, const float *dout, const int32_t *idx, float *dY,
                                               pdgn_stream_t stream) {
    if (!wgs_ok(b, n, k, ldy, T, P, C, off, offc)) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    long long total = (long long)b * n * P * C;
    hipLaunchKernelGGL(wgs_bwd_kernel, dim3(cdiv(total, WGS_THREADS)), dim3(WGS_THREADS), 0,
                       (hipStream_t)stream, total, n, k, ldy, T, P, C, off, offc, dout, idx, dY);
    return pdgn_launch_status();
}
