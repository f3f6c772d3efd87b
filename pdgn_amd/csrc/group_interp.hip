// group_interp.hip -- neighbour grouping and 3-NN feature interpolation (forward + backward).
//
// Semantics: lib/pointops/src/grouping/grouping_cuda_kernel.cu:28-46,60-74 and
// interpolation/interpolation_cuda_kernel.cu:90-114,181-195 of the reference.
//
// These are HBM/L2-bound gathers and scatters.  Forward: one thread per output (point, slot)
// walks the channels, so idx/weight are read once and every store is a coalesced row segment.
// Backward: instead of the reference's one-workgroup-per-batch global float atomics, each
// workgroup accumulates a (channel-tile x n) slab of grad_points in LDS (ds_add_f32) over its
// share of the (point, slot) pairs and flushes it with contiguous global atomics.
#include "common.h"

#define GI_THREADS 256
#define GI_LDS_FLOATS 16384   // 64 KiB slab

// out[b,c,j,s] = points[b,c,idx[b,j,s]]
__global__ __launch_bounds__(GI_THREADS) void grouping_fwd_kernel(
    int c, int n, int ms, int cpb, const float *__restrict__ points, const int32_t *__restrict__ idx,
    float *__restrict__ out) {
    const int bs = blockIdx.z;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;   // j * nsample + s
    if (e >= ms) return;
    const int ii = idx[(size_t)bs * ms + e];
    const int c0 = blockIdx.y * cpb;
    const int c1 = min(c, c0 + cpb);
    const float *P = points + (size_t)bs * c * n;
    float *O = out + (size_t)bs * c * ms;
    for (int ci = c0; ci < c1; ++ci) O[(size_t)ci * ms + e] = P[(size_t)ci * n + ii];
}

// grad_points[b,c,idx[b,j,s]] += grad_out[b,c,j,s]
__global__ __launch_bounds__(GI_THREADS) void grouping_bwd_kernel(
    int c, int n, int ms, int ct, int esplit, const float *__restrict__ grad_out,
    const int32_t *__restrict__ idx, float *__restrict__ grad_points) {
    __shared__ float slab[GI_LDS_FLOATS];
    const int bs = blockIdx.z;
    const int c0 = blockIdx.y * ct;
    const int nc = min(ct, c - c0);
    for (int i = threadIdx.x; i < nc * n; i += GI_THREADS) slab[i] = 0.f;
    __syncthreads();
    const int per = (ms + esplit - 1) / esplit;
    const int e0 = blockIdx.x * per, e1 = min(ms, e0 + per);
    const float *G = grad_out + ((size_t)bs * c + c0) * ms;
    const int32_t *I = idx + (size_t)bs * ms;
    for (int e = e0 + threadIdx.x; e < e1; e += GI_THREADS) {
        const int ii = I[e];
        for (int ci = 0; ci < nc; ++ci) atomicAdd(&slab[ci * n + ii], G[(size_t)ci * ms + e]);
    }
    __syncthreads();
    float *D = grad_points + ((size_t)bs * c + c0) * n;
    for (int i = threadIdx.x; i < nc * n; i += GI_THREADS) {
        float v = slab[i];
        if (v != 0.f) atomicAdd(&D[i], v);
    }
}

// Fallback when one channel row does not fit the LDS slab (n > 16384): plain global atomics.
__global__ __launch_bounds__(GI_THREADS) void grouping_bwd_global_kernel(
    int c, int n, int ms, const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
    float *__restrict__ grad_points) {
    const int bs = blockIdx.z, ci = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ms) return;
    atomicAdd(&grad_points[((size_t)bs * c + ci) * n + idx[(size_t)bs * ms + e]],
              grad_out[((size_t)bs * c + ci) * ms + e]);
}

// out[b,c,j] = w0*p[i0] + w1*p[i1] + w2*p[i2]   (fma(w2,p2, fma(w1,p1, w0*p0)), :194 contracted)
__global__ __launch_bounds__(GI_THREADS) void interp_fwd_kernel(
    int c, int m, int n, int cpb, const float *__restrict__ points, const int32_t *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out) {
    const int bs = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const size_t o3 = ((size_t)bs * n + j) * 3;
    const int i0 = idx[o3], i1 = idx[o3 + 1], i2 = idx[o3 + 2];
    const float w0 = weight[o3], w1 = weight[o3 + 1], w2 = weight[o3 + 2];
    const int c0 = blockIdx.y * cpb, c1 = min(c, c0 + cpb);
    for (int ci = c0; ci < c1; ++ci) {
        const float *P = points + ((size_t)bs * c + ci) * m;
        out[((size_t)bs * c + ci) * n + j] =
            __fmaf_rn(w2, P[i2], __fmaf_rn(w1, P[i1], __fmul_rn(w0, P[i0])));
    }
}

// grad_points[b,c,i_t] += grad_out[b,c,j] * w_t, t = 0..2
__global__ __launch_bounds__(GI_THREADS) void interp_bwd_kernel(
    int c, int n, int m, int ct, int esplit, const float *__restrict__ grad_out,
    const int32_t *__restrict__ idx, const float *__restrict__ weight, float *__restrict__ grad_points) {
    __shared__ float slab[GI_LDS_FLOATS];
    const int bs = blockIdx.z;
    const int c0 = blockIdx.y * ct;
    const int nc = min(ct, c - c0);
    for (int i = threadIdx.x; i < nc * m; i += GI_THREADS) slab[i] = 0.f;
    __syncthreads();
    const int per = (n + esplit - 1) / esplit;
    const int j0 = blockIdx.x * per, j1 = min(n, j0 + per);
    for (int j = j0 + threadIdx.x; j < j1; j += GI_THREADS) {
        const size_t o3 = ((size_t)bs * n + j) * 3;
        const int i0 = idx[o3], i1 = idx[o3 + 1], i2 = idx[o3 + 2];
        const float w0 = weight[o3], w1 = weight[o3 + 1], w2 = weight[o3 + 2];
        for (int ci = 0; ci < nc; ++ci) {
            float g = grad_out[((size_t)bs * c + c0 + ci) * n + j];
            atomicAdd(&slab[ci * m + i0], g * w0);
            atomicAdd(&slab[ci * m + i1], g * w1);
            atomicAdd(&slab[ci * m + i2], g * w2);
        }
    }
    __syncthreads();
    float *D = grad_points + ((size_t)bs * c + c0) * m;
    for (int i = threadIdx.x; i < nc * m; i += GI_THREADS) {
        float v = slab[i];
        if (v != 0.f) atomicAdd(&D[i], v);
    }
}

__global__ __launch_bounds__(GI_THREADS) void interp_bwd_global_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
    const int bs = blockIdx.z, ci = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const size_t o3 = ((size_t)bs * n + j) * 3;
    float g = grad_out[((size_t)bs * c + ci) * n + j];
    float *D = grad_points + ((size_t)bs * c + ci) * m;
    for (int t = 0; t < 3; ++t) atomicAdd(&D[idx[o3 + t]], g * weight[o3 + t]);
}

static bool dims_ok(int b, int c, int x, int y) {
    return b >= 0 && c >= 0 && x >= 0 && y >= 0 && b <= 65535 && c <= 65535 * 64;
}

// channels per workgroup in the forward kernels: keep >= ~2048 workgroups when possible.
static int fwd_cpb(int c, long long blocks_per_channel_pass) {
    int cpb = c;
    while (cpb > 1 && blocks_per_channel_pass * ((c + cpb - 1) / cpb) < 2048) cpb = (cpb + 1) / 2;
    return cpb;
}

extern "C" int pdgn_grouping_forward(int b, int c, int n, int m, int nsample, const float *points,
                                     const int32_t *idx, float *out, pdgn_stream_t stream) {
    if (!dims_ok(b, c, n, m) || nsample < 0) return PDGN_ERR_INVALID;
    const long long ms = (long long)m * nsample;
    if (b == 0 || c == 0 || ms == 0) return 0;
    if (ms > 0x7fffffffLL) return PDGN_ERR_INVALID;
    const int gx = cdiv(ms, GI_THREADS);
    const int cpb = fwd_cpb(c, (long long)gx * b);
    dim3 grid(gx, cdiv(c, cpb), b);
    hipLaunchKernelGGL(grouping_fwd_kernel, grid, dim3(GI_THREADS), 0, (hipStream_t)stream, c, n,
                       (int)ms, cpb, points, idx, out);
    return pdgn_launch_status();
}

extern "C" int pdgn_grouping_backward(int b, int c, int n, int m, int nsample, const float *grad_out,
                                      const int32_t *idx, float *grad_points, pdgn_stream_t stream) {
    if (!dims_ok(b, c, n, m) || nsample < 0) return PDGN_ERR_INVALID;
    const long long ms = (long long)m * nsample;
    if (b == 0 || c == 0 || ms == 0 || n == 0) return 0;
    if (ms > 0x7fffffffLL) return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (n <= GI_LDS_FLOATS) {
        int ct = GI_LDS_FLOATS / n;
        if (ct > c) ct = c;
        const int gy = cdiv(c, ct);
        int esplit = (int)(2048 / ((long long)gy * b));
        const int max_split = cdiv(ms, 4 * GI_THREADS);
        esplit = esplit < 1 ? 1 : (esplit > max_split ? max_split : esplit);
        dim3 grid(esplit, gy, b);
        hipLaunchKernelGGL(grouping_bwd_kernel, grid, dim3(GI_THREADS), 0, s, c, n, (int)ms, ct, esplit,
                           grad_out, idx, grad_points);
    } else {
        dim3 grid(cdiv(ms, GI_THREADS), c, b);
        hipLaunchKernelGGL(grouping_bwd_global_kernel, grid, dim3(GI_THREADS), 0, s, c, n, (int)ms,
                           grad_out, idx, grad_points);
    }
    return pdgn_launch_status();
}

extern "C" int pdgn_interpolation_forward(int b, int c, int m, int n, const float *points,
                                          const int32_t *idx, const float *weight, float *out,
                                          pdgn_stream_t stream) {
    if (!dims_ok(b, c, n, m)) return PDGN_ERR_INVALID;
    if (b == 0 || c == 0 || n == 0) return 0;
    const int gx = cdiv(n, GI_THREADS);
    const int cpb = fwd_cpb(c, (long long)gx * b);
    dim3 grid(gx, cdiv(c, cpb), b);
    hipLaunchKernelGGL(interp_fwd_kernel, grid, dim3(GI_THREADS), 0, (hipStream_t)stream, c, m, n, cpb,
                       points, idx, weight, out);
    return pdgn_launch_status();
}

extern "C" int pdgn_interpolation_backward(int b, int c, int n, int m, const float *grad_out,
                                           const int32_t *idx, const float *weight,
                                           float *grad_points, pdgn_stream_t stream) {
    if (!dims_ok(b, c, n, m)) return PDGN_ERR_INVALID;
    if (b == 0 || c == 0 || n == 0 || m == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (m <= GI_LDS_FLOATS) {
        int ct = GI_LDS_FLOATS / m;
        if (ct > c) ct = c;
        const int gy = cdiv(c, ct);
        int esplit = (int)(2048 / ((long long)gy * b));
        const int max_split = cdiv(n, 4 * GI_THREADS);
        esplit = esplit < 1 ? 1 : (esplit > max_split ? max_split : esplit);
        dim3 grid(esplit, gy, b);
        hipLaunchKernelGGL(interp_bwd_kernel, grid, dim3(GI_THREADS), 0, s, c, n, m, ct, esplit, grad_out,
                           idx, weight, grad_points);
    } else {
        dim3 grid(cdiv(n, GI_THREADS), c, b);
        hipLaunchKernelGGL(interp_bwd_global_kernel, grid, dim3(GI_THREADS), 0, s, c, n, m, grad_out, idx,
                           weight, grad_points);
    }
    return pdgn_launch_status();
}
