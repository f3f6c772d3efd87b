// pointops_extra.hip -- the pointops entry points PDGN itself never calls, for
// drop-in completeness with PointWeb-style callers of lib/pointops: ball query, farthest point sampling,
// gathering (+adjoint), integer grouping, feature distribute / gather, label statistics.
// All scans stage the candidate cloud through LDS in coalesced tiles; distances use the contraction-exact
// chain of common.h (what nvcc makes of the reference's expressions).
#include "common.h"

#define PX_THREADS 256
#define PX_TILE 1024

// Cooperative load of points [t0, t0+tn) of one cloud into LDS (x,y,z interleaved as in memory).
__device__ __forceinline__ void px_stage(const float *__restrict__ P, int t0, int tn, float *__restrict__ s) {
    for (int i = threadIdx.x; i < tn * 3; i += blockDim.x) s[i] = P[(size_t)t0 * 3 + i];
}

// ballquery_cuda_kernel_fast (ballquery/ballquery_cuda_kernel.cu:47-80): one thread per query; the first hit fills
// every slot, later hits overwrite slots 1.. in index order; an empty ball leaves the caller's zeros.
// WITH_STAT: labelstat_and_ballquery (labelstat_cuda_kernel.cu:6-45) -- also sums label_stat over the kept hits.
template <bool WITH_STAT>
__global__ __launch_bounds__(PX_THREADS) void ballquery_kernel(int n, int m, float radius, int nsample, int nclass,
                                                               const float *__restrict__ new_xyz,
                                                               const float *__restrict__ xyz,
                                                               const int32_t *__restrict__ label_stat,
                                                               int32_t *__restrict__ idx, int32_t *__restrict__ new_stat) {
    __shared__ float tile[PX_TILE * 3];
    const int bs = blockIdx.y, q = blockIdx.x * PX_THREADS + threadIdx.x;
    const bool live = q < m;
    const float *P = xyz + (size_t)bs * n * 3;
    const int32_t *LS = WITH_STAT ? label_stat + (size_t)bs * n * nclass : nullptr;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    int32_t *out = nullptr, *st = nullptr;
    if (live) {
        const float *Q = new_xyz + ((size_t)bs * m + q) * 3;
        qx = Q[0]; qy = Q[1]; qz = Q[2];
        out = idx + ((size_t)bs * m + q) * nsample;
        if (WITH_STAT) {
            st = new_stat + ((size_t)bs * m + q) * nclass;
            for (int i = 0; i < nclass; ++i) st[i] = 0;
        }
    }
    const float r2 = radius * radius;
    int cnt = live ? 0 : nsample;
    for (int t0 = 0; t0 < n; t0 += PX_TILE) {
        const int tn = min(PX_TILE, n - t0);
        __syncthreads();
        px_stage(P, t0, tn, tile);
        __syncthreads();
        for (int k = 0; k < tn && cnt < nsample; ++k) {
            const float d2 = sqdist3(qx, qy, qz, tile[3 * k], tile[3 * k + 1], tile[3 * k + 2]);
            if (d2 < r2) {
                const int g = t0 + k;
                if (WITH_STAT)
                    for (int i = 0; i < nclass; ++i) st[i] += LS[(size_t)g * nclass + i];
                if (cnt == 0)
                    for (int l = 0; l < nsample; ++l) out[l] = g;
                out[cnt] = g;
                ++cnt;
            }
        }
        if (__syncthreads_and(cnt >= nsample)) break;
    }
}

// labelstat_ballrange (labelstat_cuda_kernel.cu:66-95): label histogram over EVERY point of the ball.
__global__ __launch_bounds__(PX_THREADS) void labelstat_ballrange_kernel(int n, int m, float radius, int nclass,
                                                                         const float *__restrict__ new_xyz,
                                                                         const float *__restrict__ xyz,
                                                                         const int32_t *__restrict__ label_stat,
                                                                         int32_t *__restrict__ new_stat) {
    __shared__ float tile[PX_TILE * 3];
    const int bs = blockIdx.y, q = blockIdx.x * PX_THREADS + threadIdx.x;
    const bool live = q < m;
    const float *P = xyz + (size_t)bs * n * 3;
    const int32_t *LS = label_stat + (size_t)bs * n * nclass;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    int32_t *st = nullptr;
    if (live) {
        const float *Q = new_xyz + ((size_t)bs * m + q) * 3;
        qx = Q[0]; qy = Q[1]; qz = Q[2];
        st = new_stat + ((size_t)bs * m + q) * nclass;
        for (int i = 0; i < nclass; ++i) st[i] = 0;
    }
    const float r2 = radius * radius;
    for (int t0 = 0; t0 < n; t0 += PX_TILE) {
        const int tn = min(PX_TILE, n - t0);
        __syncthreads();
        px_stage(P, t0, tn, tile);
        __syncthreads();
        if (live)
            for (int k = 0; k < tn; ++k)
                if (sqdist3(qx, qy, qz, tile[3 * k], tile[3 * k + 1], tile[3 * k + 2]) < r2)
                    for (int i = 0; i < nclass; ++i) st[i] += LS[(size_t)(t0 + k) * nclass + i];
    }
}

// labelstat_idx (labelstat_cuda_kernel.cu:118-140): new_stat[b,j,:] = sum_s label_stat[b, idx[b,j,s], :]; one thread
// per (query, class).
__global__ void labelstat_idx_kernel(int n, int m, int nsample, int nclass, const int32_t *__restrict__ label_stat,
                                     const int32_t *__restrict__ idx, int32_t *__restrict__ new_stat) {
    const int bs = blockIdx.y;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)m * nclass) return;
    const int q = (int)(e / nclass), c = (int)(e % nclass);
    const int32_t *I = idx + ((size_t)bs * m + q) * nsample;
    const int32_t *LS = label_stat + (size_t)bs * n * nclass;
    int acc = 0;
    for (int s = 0; s < nsample; ++s) acc += LS[(size_t)I[s] * nclass + c];
    new_stat[((size_t)bs * m + q) * nclass + c] = acc;
}

// featuredistribute (featuredistribute_cuda_kernel.cu:4-31): for every xyz point the index of the nearest max_xyz
// point; strict <, initial (100000, -1) as in the reference.
__global__ __launch_bounds__(PX_THREADS) void featuredistribute_kernel(int n, int m, const float *__restrict__ max_xyz,
                                                                       const float *__restrict__ xyz,
                                                                       int32_t *__restrict__ out) {
    __shared__ float tile[PX_TILE * 3];
    const int bs = blockIdx.y, q = blockIdx.x * PX_THREADS + threadIdx.x;
    const bool live = q < m;
    const float *P = max_xyz + (size_t)bs * n * 3;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        const float *Q = xyz + ((size_t)bs * m + q) * 3;
        qx = Q[0]; qy = Q[1]; qz = Q[2];
    }
    float best = 100000.f;
    int besti = -1;
    for (int t0 = 0; t0 < n; t0 += PX_TILE) {
        const int tn = min(PX_TILE, n - t0);
        __syncthreads();
        px_stage(P, t0, tn, tile);
        __syncthreads();
        for (int k = 0; k < tn; ++k) {
            const float d2 = sqdist3(tile[3 * k], tile[3 * k + 1], tile[3 * k + 2], qx, qy, qz);
            if (d2 < best) { best = d2; besti = t0 + k; }
        }
    }
    if (live) out[(size_t)bs * m + q] = besti;
}

// gathering (sampling_cuda_kernel.cu:6-36) == featuregather (featuredistribute_cuda_kernel.cu:52-106):
// out[b,c,j] = points[b,c,idx[b,j]]; adjoint scatters with float atomics into a zeroed buffer.
template <typename T>
__global__ void gather_fwd_kernel(int c, int n, int m, const T *__restrict__ points, const int32_t *__restrict__ idx,
                                  T *__restrict__ out) {
    const int bs = blockIdx.z, ch = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    out[((size_t)bs * c + ch) * m + j] = points[((size_t)bs * c + ch) * n + idx[(size_t)bs * m + j]];
}

__global__ void gather_bwd_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                  const int32_t *__restrict__ idx, float *__restrict__ grad_points) {
    const int bs = blockIdx.z, ch = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    atomicAdd(grad_points + ((size_t)bs * c + ch) * n + idx[(size_t)bs * m + j], grad_out[((size_t)bs * c + ch) * m + j]);
}

// furthestsampling (sampling_cuda_kernel.cu:59-168): one workgroup per cloud; idx[0] = 0, then m-1 rounds of
// "temp = min(temp, d(last, .)); last = argmax temp".  The reference's tree reduction breaks exact ties by thread
// slot; here ties go to the LOWEST index (deterministic; identical whenever the maximum is unique).
#define FPS_THREADS 1024
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(int n, int m, const float *__restrict__ xyz,
                                                          float *__restrict__ temp, int32_t *__restrict__ idxs) {
    __shared__ float sd[FPS_THREADS / PDGN_WAVE];
    __shared__ int si[FPS_THREADS / PDGN_WAVE];
    __shared__ int s_old;
    const int bs = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *P = xyz + (size_t)bs * n * 3;
    float *T = temp + (size_t)bs * n;
    int32_t *O = idxs + (size_t)bs * m;
    int old = 0;
    if (tid == 0) O[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = P[old * 3], y1 = P[old * 3 + 1], z1 = P[old * 3 + 2];
        float best = -1.f;
        int besti = 0;
        for (int k = tid; k < n; k += FPS_THREADS) {
            const float d = sqdist3(P[k * 3], P[k * 3 + 1], P[k * 3 + 2], x1, y1, z1);
            const float d2 = fminf(d, T[k]);
            T[k] = d2;
            if (d2 > best) { best = d2; besti = k; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(besti, off, 64);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (lane == 0) { sd[wave] = best; si[wave] = besti; }
        __syncthreads();
        if (wave == 0) {
            float b2 = lane < FPS_THREADS / PDGN_WAVE ? sd[lane] : -2.f;
            int i2 = lane < FPS_THREADS / PDGN_WAVE ? si[lane] : 0x7fffffff;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const float ob = __shfl_xor(b2, off, 64);
                const int oi = __shfl_xor(i2, off, 64);
                if (ob > b2 || (ob == b2 && oi < i2)) { b2 = ob; i2 = oi; }
            }
            if (lane == 0) { s_old = i2; O[j] = i2; }
        }
        __syncthreads();
        old = s_old;
    }
}

// ---------------------------------------------------------------------------- C ABI
extern "C" int pdgn_ballquery(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                              int32_t *idx, pdgn_stream_t stream) {
    if (b < 0 || n < 1 || m < 1 || nsample < 1 || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(ballquery_kernel<false>, dim3(cdiv(m, PX_THREADS), b), dim3(PX_THREADS), 0, (hipStream_t)stream, n, m,
                       radius, nsample, 0, new_xyz, xyz, (const int32_t *)nullptr, idx, (int32_t *)nullptr);
    return pdgn_launch_status();
}

extern "C" int pdgn_labelstat_and_ballquery(int b, int n, int m, float radius, int nsample, int nclass,
                                            const float *new_xyz, const float *xyz, const int32_t *label_stat,
                                            int32_t *idx, int32_t *new_label_stat, pdgn_stream_t stream) {
    if (b < 0 || n < 1 || m < 1 || nsample < 1 || nclass < 1 || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(ballquery_kernel<true>, dim3(cdiv(m, PX_THREADS), b), dim3(PX_THREADS), 0, (hipStream_t)stream, n, m,
                       radius, nsample, nclass, new_xyz, xyz, label_stat, idx, new_label_stat);
    return pdgn_launch_status();
}

extern "C" int pdgn_labelstat_ballrange(int b, int n, int m, float radius, int nclass, const float *new_xyz,
                                        const float *xyz, const int32_t *label_stat, int32_t *new_label_stat,
                                        pdgn_stream_t stream) {
    if (b < 0 || n < 1 || m < 1 || nclass < 1 || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(labelstat_ballrange_kernel, dim3(cdiv(m, PX_THREADS), b), dim3(PX_THREADS), 0, (hipStream_t)stream, n,
                       m, radius, nclass, new_xyz, xyz, label_stat, new_label_stat);
    return pdgn_launch_status();
}

extern "C" int pdgn_labelstat_idx(int b, int n, int m, int nsample, int nclass, const int32_t *label_stat,
                                  const int32_t *idx, int32_t *new_label_stat, pdgn_stream_t stream) {
    if (b < 0 || n < 1 || m < 1 || nsample < 1 || nclass < 1 || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(labelstat_idx_kernel, dim3(cdiv((long long)m * nclass, 256), b), dim3(256), 0, (hipStream_t)stream, n, m,
                       nsample, nclass, label_stat, idx, new_label_stat);
    return pdgn_launch_status();
}

extern "C" int pdgn_featuredistribute(int b, int n, int m, const float *max_xyz, const float *xyz,
                                      int32_t *distribute_idx, pdgn_stream_t stream) {
    if (b < 0 || n < 1 || m < 1 || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(featuredistribute_kernel, dim3(cdiv(m, PX_THREADS), b), dim3(PX_THREADS), 0, (hipStream_t)stream, n, m,
                       max_xyz, xyz, distribute_idx);
    return pdgn_launch_status();
}

extern "C" int pdgn_gathering_forward(int b, int c, int n, int m, const float *points, const int32_t *idx, float *out,
                                      pdgn_stream_t stream) {
    if (b < 0 || c < 1 || n < 1 || m < 1 || b > 65535 || c > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(gather_fwd_kernel<float>, dim3(cdiv(m, 256), c, b), dim3(256), 0, (hipStream_t)stream, c, n, m, points,
                       idx, out);
    return pdgn_launch_status();
}

extern "C" int pdgn_gathering_backward(int b, int c, int n, int m, const float *grad_out, const int32_t *idx,
                                       float *grad_points, pdgn_stream_t stream) {
    if (b < 0 || c < 1 || n < 1 || m < 1 || b > 65535 || c > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(gather_bwd_kernel, dim3(cdiv(m, 256), c, b), dim3(256), 0, (hipStream_t)stream, c, n, m, grad_out, idx,
                       grad_points);
    return pdgn_launch_status();
}

// grouping_int (grouping_int_cuda_kernel.cu): int64 features (b,c,n), idx (b,m,ns) -> (b,c,m,ns); a gather with
// m*ns outputs per channel.
extern "C" int pdgn_grouping_int_forward(int b, int c, int n, int m, int nsample, const long long *points,
                                         const int32_t *idx, long long *out, pdgn_stream_t stream) {
    if (b < 0 || c < 1 || n < 1 || m < 1 || nsample < 1 || b > 65535 || c > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(gather_fwd_kernel<long long>, dim3(cdiv((long long)m * nsample, 256), c, b), dim3(256), 0,
                       (hipStream_t)stream, c, n, m * nsample, points, idx, out);
    return pdgn_launch_status();
}

extern "C" int pdgn_furthestsampling(int b, int n, int m, const float *xyz, float *temp, int32_t *idx,
                                     pdgn_stream_t stream) {
    if (b < 0 || n < 1 || m < 1) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(fps_kernel, dim3(b), dim3(FPS_THREADS), 0, (hipStream_t)stream, n, m, xyz, temp, idx);
    return pdgn_launch_status();
}
