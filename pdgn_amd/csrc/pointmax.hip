// pointmax.hip -- MaxPool2d((1, N)) over the points of a point-major (B, N, C) tensor, with the argmax the adjoint needs.
//
// The generator's blocks open with it (models/PDGNet_v2.py:699/:736/:777/:810: xs = self.maxpool(x)).  torch's
// max(dim=1) on these shapes is a 50-110 us reduction (one workgroup row per output, eight launches a step = 0.5 ms on
// the default stream); here the points are split over the grid (lane = channel, waves interleave rows, eight loads in
// flight), partial maxima go to a scratch and a second tiny kernel joins them.  Ties go to the lowest point index.
#include "common.h"

#define PM_THREADS 256
#define PM_ROWS 128            // points per workgroup

__global__ __launch_bounds__(PM_THREADS) void pointmax_partial_kernel(int n, int c, int ns, const float *__restrict__ x,
                                                                      float *__restrict__ pval, int32_t *__restrict__ parg) {
    __shared__ float sv[PM_THREADS];
    __shared__ int si[PM_THREADS];
    const int b = blockIdx.z, s = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + lane;
    const int r0 = s * PM_ROWS, r1 = min(n, r0 + PM_ROWS);
    float best = -INFINITY;
    int arg = r0;
    if (ch < c) {
        const float *X = x + ((size_t)b * n) * c + ch;
        for (int r = r0 + w; r < r1; r += 4 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (r + 4 * u < r1) ? X[(size_t)(r + 4 * u) * c] : -INFINITY;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v[u] > best) { best = v[u]; arg = r + 4 * u; }      // rows ascend within a thread: first maximum kept
        }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = arg;
    __syncthreads();
    if (w == 0 && ch < c) {
        for (int j = 1; j < 4; ++j) {
            const float v = sv[j * 64 + lane];
            const int a = si[j * 64 + lane];
            if (v > best || (v == best && a < arg)) { best = v; arg = a; }
        }
        pval[((size_t)b * ns + s) * c + ch] = best;
        parg[((size_t)b * ns + s) * c + ch] = arg;
    }
}

__global__ __launch_bounds__(PM_THREADS) void pointmax_join_kernel(long long total, int c, int ns, const float *__restrict__ pval,
                                                                   const int32_t *__restrict__ parg, float *__restrict__ out,
                                                                   int32_t *__restrict__ arg) {
    const long long e = (long long)blockIdx.x * PM_THREADS + threadIdx.x;      // b * c + ch
    if (e >= total) return;
    const long long b = e / c;
    const int ch = (int)(e - b * c);
    float best = -INFINITY;
    int a = 0;
    for (int s = 0; s < ns; ++s) {                             // chunks ascend: a strict > keeps the lowest index
        const float v = pval[(b * ns + s) * c + ch];
        if (s == 0 || v > best) { best = v; a = parg[(b * ns + s) * c + ch]; }
    }
    out[e] = best;
    arg[e] = a;
}

// dx[b, r, ch] = (r == arg[b, ch]) ? g[b, ch] : 0 -- the whole dense gradient in one pass (torch: zero-fill + scatter)
__global__ __launch_bounds__(PM_THREADS) void pointmax_bwd_kernel(long long total4, int n, int c4, const float *__restrict__ g,
                                                                  const int32_t *__restrict__ arg, float *__restrict__ dx) {
    const long long e = (long long)blockIdx.x * PM_THREADS + threadIdx.x;      // (b * n + r) * c4 + cv
    if (e >= total4) return;
    const int cv = (int)(e % c4);
    const long long br = e / c4;
    const long long b = br / n;
    const int r = (int)(br - b * n);
    const float4 gv = *reinterpret_cast<const float4 *>(g + (b * c4 + cv) * 4);
    const int4 av = *reinterpret_cast<const int4 *>(arg + (b * c4 + cv) * 4);
    float4 o;
    o.x = av.x == r ? gv.x : 0.f; o.y = av.y == r ? gv.y : 0.f; o.z = av.z == r ? gv.z : 0.f; o.w = av.w == r ? gv.w : 0.f;
    *reinterpret_cast<float4 *>(dx + e * 4) = o;
}

extern "C" long long pdgn_point_max_scratch(int b, int n, int c) {
    if (b < 1 || n < 1 || c < 1) return PDGN_ERR_INVALID;
    return (long long)b * cdiv(n, PM_ROWS) * c;                // floats, and as many int32
}

extern "C" int pdgn_point_max(int b, int n, int c, const float *x, float *scratch_val, int32_t *scratch_arg, float *out,
                              int32_t *arg, pdgn_stream_t stream) {
    if (b < 1 || n < 1 || c < 1 || b > 65535) return PDGN_ERR_INVALID;
    const int ns = cdiv(n, PM_ROWS);
    if (ns > 65535) return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pointmax_partial_kernel, dim3(cdiv(c, 64), ns, b), dim3(PM_THREADS), 0, s, n, c, ns, x, scratch_val,
                       scratch_arg);
    const long long total = (long long)b * c;
    hipLaunchKernelGGL(pointmax_join_kernel, dim3(cdiv(total, PM_THREADS)), dim3(PM_THREADS), 0, s, total, c, ns, scratch_val,
                       scratch_arg, out, arg);
    return pdgn_launch_status();
}

extern "C" int pdgn_point_max_backward(int b, int n, int c, const float *grad_out, const int32_t *arg, float *grad_x,
                                       pdgn_stream_t stream) {
    if (b < 1 || n < 1 || c < 4 || c % 4) return PDGN_ERR_INVALID;
    const long long total4 = (long long)b * n * (c / 4);
    hipLaunchKernelGGL(pointmax_bwd_kernel, dim3(cdiv(total4, PM_THREADS)), dim3(PM_THREADS), 0, (hipStream_t)stream, total4, n,
                       c / 4, grad_out, arg, grad_x);
    return pdgn_launch_status();
}
