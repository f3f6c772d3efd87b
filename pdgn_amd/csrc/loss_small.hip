// loss_small.hip -- the scalar ends of the step's losses as single launches.
//
// The adversarial terms are nn.MSELoss against a constant (models/PDGNet_v2.py:186-190, 246-250: mse(D(x), 1) / mse(D(x), 0)
// on a (B, 1) score), the local-pair terms sums of Chamfer minima (utils/chamfer_loss.py:16-20).  As torch ops each is two
// launches forward (elementwise + reduction, or reduction + scale) and two backward (a zero-filled or expanded gradient + the
// elementwise adjoint) -- 24 such terms per iteration, every launch ~10 us of host time on the issuing thread.  One workgroup
// each way; the sums run in a fixed order (deterministic).
#include "common.h"

#define LS_THREADS 1024

__device__ __forceinline__ float ls_block_sum(float v) {
    __shared__ float red[LS_THREADS / PDGN_WAVE];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane_id() == 0) red[threadIdx.x / PDGN_WAVE] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < LS_THREADS / PDGN_WAVE; ++w) s += red[w];
    return s;                                                      // valid in thread 0
}

// out[0] = scale * sum x
__global__ __launch_bounds__(LS_THREADS) void scaled_sum_kernel(long long n, const float *__restrict__ x, float scale,
                                                               float *__restrict__ out) {
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += LS_THREADS) s += x[i];
    s = ls_block_sum(s);
    if (threadIdx.x == 0) out[0] = scale * s;
}

// out[0] = scale * mean (x - t)^2
__global__ __launch_bounds__(LS_THREADS) void mse_const_fwd_kernel(long long n, const float *__restrict__ x, float t, float scale,
                                                                  float *__restrict__ out) {
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += LS_THREADS) {
        const float d = x[i] - t;
        s = __fmaf_rn(d, d, s);
    }
    s = ls_block_sum(s);
    if (threadIdx.x == 0) out[0] = scale * (s / (float)n);
}

// dx = g[0] * scale * 2 (x - t) / n
__global__ __launch_bounds__(256) void mse_const_bwd_kernel(long long n, const float *__restrict__ x, float t, float scale,
                                                            const float *__restrict__ g, float *__restrict__ dx) {
    const float c = g[0] * scale * 2.0f / (float)n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dx[i] = c * (x[i] - t);
}

extern "C" int pdgn_scaled_sum(long long n, const float *x, float scale, float *out, pdgn_stream_t stream) {
    if (n < 0) return PDGN_ERR_INVALID;
    hipLaunchKernelGGL(scaled_sum_kernel, dim3(1), dim3(LS_THREADS), 0, (hipStream_t)stream, n, x, scale, out);
    return pdgn_launch_status();
}

extern "C" int pdgn_mse_const(long long n, const float *x, float target, float scale, float *out, pdgn_stream_t stream) {
    if (n < 1) return PDGN_ERR_INVALID;
    hipLaunchKernelGGL(mse_const_fwd_kernel, dim3(1), dim3(LS_THREADS), 0, (hipStream_t)stream, n, x, target, scale, out);
    return pdgn_launch_status();
}

extern "C" int pdgn_mse_const_backward(long long n, const float *x, float target, float scale, const float *g, float *dx,
                                       pdgn_stream_t stream) {
    if (n < 1) return PDGN_ERR_INVALID;
    const int grid = (int)(n < 256 * 64 ? cdiv(n, 256) : 64);
    hipLaunchKernelGGL(mse_const_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, x, target, scale, g, dx);
    return pdgn_launch_status();
}
