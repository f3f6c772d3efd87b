// localpair.hip -- the shape-preserving "local pair" loss of the PDGN generator step
// (models/PDGNet_v2.py:127-155) as two fused kernels instead of grouping + transposes + bmm.
//
//   local_stats : for every query point, gather its K neighbours (indices from pdgn_knnquery) and
//                 emit the neighbourhood mean (3) and covariance (3x3) -- the reference's
//                 grouping -> view(-1,3,20) -> compute_mean_covariance (:127-134, :142-147) without
//                 the (B,3,M,K) grouped tensor or the (B*M,3,K)x(B*M,K,3) bmm.
//   chamfer_gram: utils/chamfer_loss.py:13-38 -- P[i,j] = (|x_i|^2 + |y_j|^2) - 2<x_i,y_j> in
//                 the reference's Gram form (no clamp), row minima and column minima with
//                 argmins, both directions in one launch; the (B,M,N) matrix never exists.
// Both have hand-written adjoints (scatter through the argmins / neighbour indices).
#include "common.h"
#include <stdlib.h>

#define LP_THREADS 256
#define LP_LDS_FLOATS 8192       // 32 KiB slab for the privatised scatter (n <= 2730)
#define LP_PTS_FLOATS 6144       // 24 KiB: the sample's points staged for the adjoint's gathers (n <= 2048)

// ---------------------------------------------------------------------------- local statistics
// xyz (b,n,3), idx (b,m,K) -> mu (b,m,3), cov (b,m,9);  cov = (1/K) sum_s t_s t_s^T, t_s = p_s - mu
__global__ __launch_bounds__(LP_THREADS) void local_stats_fwd_kernel(
    int n, int m, int K, const float *__restrict__ xyz, const int32_t *__restrict__ idx,
    float *__restrict__ mu, float *__restrict__ cov) {
    const int bs = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const float *P = xyz + (size_t)bs * n * 3;
    const int32_t *I = idx + ((size_t)bs * m + q) * K;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int s = 0; s < K; ++s) {
        const float *p = P + (size_t)I[s] * 3;
        sx += p[0]; sy += p[1]; sz += p[2];
    }
    const float invK = 1.0f / (float)K;
    const float mx = sx * invK, my = sy * invK, mz = sz * invK;
    float c00 = 0.f, c01 = 0.f, c02 = 0.f, c11 = 0.f, c12 = 0.f, c22 = 0.f;
    for (int s = 0; s < K; ++s) {
        const float *p = P + (size_t)I[s] * 3;
        const float tx = p[0] - mx, ty = p[1] - my, tz = p[2] - mz;
        c00 = __fmaf_rn(tx, tx, c00); c01 = __fmaf_rn(tx, ty, c01); c02 = __fmaf_rn(tx, tz, c02);
        c11 = __fmaf_rn(ty, ty, c11); c12 = __fmaf_rn(ty, tz, c12); c22 = __fmaf_rn(tz, tz, c22);
    }
    float *M = mu + ((size_t)bs * m + q) * 3;
    M[0] = mx; M[1] = my; M[2] = mz;
    float *C = cov + ((size_t)bs * m + q) * 9;
    C[0] = c00 * invK; C[1] = c01 * invK; C[2] = c02 * invK;
    C[3] = c01 * invK; C[4] = c11 * invK; C[5] = c12 * invK;
    C[6] = c02 * invK; C[7] = c12 * invK; C[8] = c22 * invK;
}

// dxyz[b, idx[q,s], :] += dmu/K + (G + G^T) t_s / K      (G = dcov of query q; sum_s t_s = 0 kills
// the path through mu).  Scatter privatised in LDS per (batch, query range), flushed with atomics.
__global__ __launch_bounds__(LP_THREADS) void local_stats_bwd_kernel(
    int n, int m, int K, int qsplit, const float *__restrict__ xyz, const int32_t *__restrict__ idx,
    const float *__restrict__ dmu, const float *__restrict__ dcov, float *__restrict__ dxyz) {
    __shared__ float slab[LP_LDS_FLOATS];
    __shared__ float pts[LP_PTS_FLOATS];                           // the sample's points: 2 K gathers per query come from LDS
    const int bs = blockIdx.y;
    const bool use_lds = n * 3 <= LP_LDS_FLOATS, use_pts = n * 3 <= LP_PTS_FLOATS;
    const float *P = xyz + (size_t)bs * n * 3;
    if (use_lds)
        for (int i = threadIdx.x; i < n * 3; i += LP_THREADS) slab[i] = 0.f;
    if (use_pts)
        for (int i = threadIdx.x; i < n * 3; i += LP_THREADS) pts[i] = P[i];
    if (use_lds || use_pts) __syncthreads();
    auto pt = [&](int j, int c) { return use_pts ? pts[j * 3 + c] : P[(size_t)j * 3 + c]; };
    float *D = dxyz + (size_t)bs * n * 3;
    const int per = (m + qsplit - 1) / qsplit;
    const int q0 = blockIdx.x * per, q1 = min(m, q0 + per);
    const float invK = 1.0f / (float)K;
    for (int q = q0 + threadIdx.x; q < q1; q += LP_THREADS) {
        const int32_t *I = idx + ((size_t)bs * m + q) * K;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int s = 0; s < K; ++s) {
            const int j = I[s];
            sx += pt(j, 0); sy += pt(j, 1); sz += pt(j, 2);
        }
        const float mx = sx * invK, my = sy * invK, mz = sz * invK;
        const float *gm = dmu + ((size_t)bs * m + q) * 3;
        const float *G = dcov + ((size_t)bs * m + q) * 9;
        const float s00 = 2.f * G[0], s01 = G[1] + G[3], s02 = G[2] + G[6];
        const float s11 = 2.f * G[4], s12 = G[5] + G[7], s22 = 2.f * G[8];
        const float bx = gm[0] * invK, by = gm[1] * invK, bz = gm[2] * invK;
        for (int s = 0; s < K; ++s) {
            const int j = I[s];
            const float tx = pt(j, 0) - mx, ty = pt(j, 1) - my, tz = pt(j, 2) - mz;
            const float gx = bx + (s00 * tx + s01 * ty + s02 * tz) * invK;
            const float gy = by + (s01 * tx + s11 * ty + s12 * tz) * invK;
            const float gz = bz + (s02 * tx + s12 * ty + s22 * tz) * invK;
            float *dst = use_lds ? &slab[j * 3] : &D[(size_t)j * 3];
            atomicAdd(dst, gx); atomicAdd(dst + 1, gy); atomicAdd(dst + 2, gz);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n * 3; i += LP_THREADS) {
            float v = slab[i];
            if (v != 0.f) atomicAdd(&D[i], v);
        }
    }
}

// ---------------------------------------------------------------------------- Chamfer (Gram form)
#define CH_MAXD 16
#define CH_TILE 1024

// direction 0: queries x (b,m,d), candidates y (b,n,d) -> minx/argx (b,m)
// direction 1: queries y, candidates x                  -> miny/argy (b,n)
// P = (r_x + r_y) - 2*dot(x,y), dot as an fma chain over d ascending: both directions evaluate
// bit-identical P[i,j].  Strict '<' scanning ascending => lowest index on ties (torch.min).
template <int D>
__global__ __launch_bounds__(LP_THREADS) void chamfer_gram_kernel(
    int m, int n, int dd, const float *__restrict__ x, const float *__restrict__ y,
    float *__restrict__ minx, int32_t *__restrict__ argx, float *__restrict__ miny,
    int32_t *__restrict__ argy, const int32_t *__restrict__ ia, const int32_t *__restrict__ ib) {
    constexpr int DS = D > 0 ? D : CH_MAXD;
    __shared__ float cand[CH_TILE * (DS + 1)];
    const int d = D > 0 ? D : dd;
    const int bs = blockIdx.y;
    const bool rev = blockIdx.z != 0;
    const int nq = rev ? n : m, nc = rev ? m : n;
    if ((int)(blockIdx.x * blockDim.x) >= nq) return;           // block-uniform
    const size_t bx = ia ? (size_t)ia[bs] : (size_t)bs, by = ib ? (size_t)ib[bs] : (size_t)bs;   // pair lists
    const float *Q = rev ? y + by * nq * d : x + bx * nq * d;
    const float *C = rev ? x + bx * nc * d : y + by * nc * d;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float qv[DS];
    float rq = 0.f;
#pragma unroll
    for (int c = 0; c < DS; ++c) {
        qv[c] = (c < d && i < nq) ? Q[(size_t)i * d + c] : 0.f;
        rq = __fmaf_rn(qv[c], qv[c], rq);
    }
    float best = INFINITY;
    int best_j = 0;
    for (int t0 = 0; t0 < nc; t0 += CH_TILE) {
        const int tn = min(CH_TILE, nc - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < tn; j += LP_THREADS) {
            float r = 0.f;
            for (int c = 0; c < d; ++c) {
                float v = C[(size_t)(t0 + j) * d + c];
                cand[j * (DS + 1) + c] = v;
                r = __fmaf_rn(v, v, r);
            }
            cand[j * (DS + 1) + DS] = r;
        }
        __syncthreads();
        for (int j = 0; j < tn; ++j) {
            const float *cv = &cand[j * (DS + 1)];
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < DS; ++c)
                if (c < d) dot = __fmaf_rn(qv[c], cv[c], dot);
            const float p = (rq + cv[DS]) - 2.0f * dot;
            const bool better = p < best;
            best = better ? p : best;
            best_j = better ? t0 + j : best_j;
        }
    }
    if (i < nq) {
        (rev ? miny : minx)[(size_t)bs * nq + i] = best;
        (rev ? argy : argx)[(size_t)bs * nq + i] = best_j;
    }
}

// d/dq of P[q, c*] = 2 (q - c*);  d/dc* = -2 (q - c*)
__global__ __launch_bounds__(LP_THREADS) void chamfer_gram_grad_kernel(
    int m, int n, int d, const float *__restrict__ x, const float *__restrict__ y,
    const float *__restrict__ gminx, const int32_t *__restrict__ argx, const float *__restrict__ gminy,
    const int32_t *__restrict__ argy, float *__restrict__ gx, float *__restrict__ gy, const float *__restrict__ guni, float gscale) {
    const int bs = blockIdx.y;
    const bool rev = blockIdx.z != 0;
    const int nq = rev ? n : m, nc = rev ? m : n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float *Q = (rev ? y : x) + ((size_t)bs * nq + i) * d;
    const int j = (rev ? argy : argx)[(size_t)bs * nq + i];
    const float *C = (rev ? x : y) + ((size_t)bs * nc + j) * d;
    // upstream gradient of the minima: per element, or (guni) ONE device scalar times gscale (the minima were summed)
    const float g = 2.0f * (guni ? guni[0] * gscale : (rev ? gminy : gminx)[(size_t)bs * nq + i]);
    float *GQ = (rev ? gy : gx) + ((size_t)bs * nq + i) * d;
    float *GC = (rev ? gx : gy) + ((size_t)bs * nc + j) * d;
    for (int c = 0; c < d; ++c) {
        const float v = g * (Q[c] - C[c]);
        atomicAdd(&GQ[c], v);
        atomicAdd(&GC[c], -v);
    }
}

// The same gradients with the sums taken in LDS: one workgroup per (sample, cloud) holds that cloud's whole gradient (nq x d
// floats), writes every point's own term 2 g (q_i - c*) there, adds the terms of the other cloud's points that chose it with LDS
// atomics (every global access a flat, coalesced sweep; the only gathers are c* and q_target) and stores the result once: no
// global atomics, no zero-fill.  35 x 1024 x 18 float atomics onto a few popular points took 56 us isolated / 85-130 us in the
// iteration for the 9-D covariance pairs.  (A per-point gather -- each thread scanning the other direction's arg-min list --
// was slower: a match is rare per lane but not per wavefront, and every one is a dependent round trip.)
#define CHL_THREADS 1024
#define CHL_MAXF 12288      // floats of one cloud's gradient in LDS (48 KB)
template <int D>
__global__ __launch_bounds__(CHL_THREADS) void chamfer_gram_grad_lds_kernel(
    int m, int n, int d, const float *__restrict__ x, const float *__restrict__ y,
    const float *__restrict__ gminx, const int32_t *__restrict__ argx, const float *__restrict__ gminy,
    const int32_t *__restrict__ argy, float *__restrict__ gx, float *__restrict__ gy, const float *__restrict__ guni, float gscale) {
    __shared__ float accs[CHL_MAXF];
    const int dd = D > 0 ? D : d;
    const int bs = blockIdx.y;
    const bool rev = blockIdx.z != 0;                              // rev: this block writes gy
    const int nq = rev ? n : m, nc = rev ? m : n;                  // own cloud / other cloud
    const float *Qb = (rev ? y : x) + (size_t)bs * nq * dd, *Cb = (rev ? x : y) + (size_t)bs * nc * dd;
    const int32_t *own_arg = (rev ? argy : argx) + (size_t)bs * nq, *oth_arg = (rev ? argx : argy) + (size_t)bs * nc;
    const float *own_g = (rev ? gminy : gminx) + (guni ? 0 : (size_t)bs * nq), *oth_g = (rev ? gminx : gminy) + (guni ? 0 : (size_t)bs * nc);
    const float uni = guni ? 2.0f * guni[0] * gscale : 0.f;
    for (int f = threadIdx.x; f < nq * dd; f += CHL_THREADS) {     // own terms: entry f = (point i, coordinate c)
        const int i = f / dd, c = f - i * dd;
        const float g = guni ? uni : 2.0f * own_g[i];
        accs[f] = g * (Qb[f] - Cb[(size_t)own_arg[i] * dd + c]);
    }
    __syncthreads();
    for (int f = threadIdx.x; f < nc * dd; f += CHL_THREADS) {     // the other cloud's points, each to the point it chose
        const int e = f / dd, c = f - e * dd;
        const int tgt = oth_arg[e];
        const float ge = guni ? uni : 2.0f * oth_g[e];
        __hip_atomic_fetch_add(&accs[tgt * dd + c], ge * (Qb[(size_t)tgt * dd + c] - Cb[f]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    float *G = (rev ? gy : gx) + (size_t)bs * nq * dd;
    for (int f = threadIdx.x; f < nq * dd; f += CHL_THREADS) G[f] = accs[f];
}

// ---------------------------------------------------------------------------- C ABI
extern "C" int pdgn_local_stats(int b, int n, int m, int k, const float *xyz, const int32_t *idx, float *mu,
                                float *cov, pdgn_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || k < 1 || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0 || m == 0) return 0;
    hipLaunchKernelGGL(local_stats_fwd_kernel, dim3(cdiv(m, LP_THREADS), b), dim3(LP_THREADS), 0,
                       (hipStream_t)stream, n, m, k, xyz, idx, mu, cov);
    return pdgn_launch_status();
}

extern "C" int pdgn_local_stats_backward(int b, int n, int m, int k, const float *xyz, const int32_t *idx,
                                         const float *dmu, const float *dcov, float *dxyz,
                                         pdgn_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || k < 1 || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0 || m == 0) return 0;
    int qsplit = 1024 / b;
    const int max_split = cdiv(m, LP_THREADS);
    qsplit = qsplit < 1 ? 1 : (qsplit > max_split ? max_split : qsplit);
    hipLaunchKernelGGL(local_stats_bwd_kernel, dim3(qsplit, b), dim3(LP_THREADS), 0, (hipStream_t)stream, n, m, k,
                       qsplit, xyz, idx, dmu, dcov, dxyz);
    return pdgn_launch_status();
}

static int chamfer_launch(dim3 grid, hipStream_t s, int m, int n, int d, const float *x, const float *y, float *minx,
                          int32_t *argx, float *miny, int32_t *argy, const int32_t *ia, const int32_t *ib) {
    if (d == 3)
        hipLaunchKernelGGL(chamfer_gram_kernel<3>, grid, dim3(LP_THREADS), 0, s, m, n, d, x, y, minx, argx, miny, argy, ia, ib);
    else if (d == 9)
        hipLaunchKernelGGL(chamfer_gram_kernel<9>, grid, dim3(LP_THREADS), 0, s, m, n, d, x, y, minx, argx, miny, argy, ia, ib);
    else
        hipLaunchKernelGGL(chamfer_gram_kernel<0>, grid, dim3(LP_THREADS), 0, s, m, n, d, x, y, minx, argx, miny, argy, ia, ib);
    return pdgn_launch_status();
}

extern "C" int pdgn_chamfer_gram(int b, int m, int n, int d, const float *x, const float *y, float *minx,
                                 int32_t *argx, float *miny, int32_t *argy, pdgn_stream_t stream) {
    if (b < 0 || m < 1 || n < 1 || d < 1 || d > CH_MAXD || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    dim3 grid(cdiv(m > n ? m : n, LP_THREADS), b, 2);
    hipStream_t s = (hipStream_t)stream;
    return chamfer_launch(grid, s, m, n, d, x, y, minx, argx, miny, argy, nullptr, nullptr);
}

// The same for `npairs` (ia[p], ib[p]) pairs of clouds drawn from x (., m, d) and y (., n, d).
extern "C" int pdgn_chamfer_gram_indexed(int npairs, int m, int n, int d, const float *x, const int32_t *ia,
                                         const float *y, const int32_t *ib, float *minx, int32_t *argx,
                                         float *miny, int32_t *argy, pdgn_stream_t stream) {
    if (npairs < 0 || m < 1 || n < 1 || d < 1 || d > CH_MAXD || npairs > 65535) return PDGN_ERR_INVALID;
    if (npairs == 0) return 0;
    dim3 grid(cdiv(m > n ? m : n, LP_THREADS), npairs, 2);
    return chamfer_launch(grid, (hipStream_t)stream, m, n, d, x, y, minx, argx, miny, argy, ia, ib);
}

static int chamfer_grad_launch(int b, int m, int n, int d, const float *x, const float *y, const float *gminx, const int32_t *argx,
                               const float *gminy, const int32_t *argy, float *gx, float *gy, const float *guni, float gscale,
                               hipStream_t s) {
    hipError_t e;
    const size_t nx = (size_t)b * m * d, ny = (size_t)b * n * d;
    static const bool in_lds = !(getenv("PDGN_CHAMFER_LDS") && getenv("PDGN_CHAMFER_LDS")[0] == '0');   // A/B switch
    if (in_lds && (size_t)m * d <= CHL_MAXF && (size_t)n * d <= CHL_MAXF) {   // sums in LDS: plain stores, nothing to clear
        dim3 grid(1, b, 2);
        if (d == 3)
            hipLaunchKernelGGL(chamfer_gram_grad_lds_kernel<3>, grid, dim3(CHL_THREADS), 0, s, m, n, d, x, y, gminx, argx, gminy, argy,
                               gx, gy, guni, gscale);
        else if (d == 9)
            hipLaunchKernelGGL(chamfer_gram_grad_lds_kernel<9>, grid, dim3(CHL_THREADS), 0, s, m, n, d, x, y, gminx, argx, gminy, argy,
                               gx, gy, guni, gscale);
        else
            hipLaunchKernelGGL(chamfer_gram_grad_lds_kernel<0>, grid, dim3(CHL_THREADS), 0, s, m, n, d, x, y, gminx, argx, gminy, argy,
                               gx, gy, guni, gscale);
        return pdgn_launch_status();
    }
    if (gy == gx + nx) {                                           // one buffer (the callers here allocate it so): one fill
        if ((e = hipMemsetAsync(gx, 0, (nx + ny) * sizeof(float), s)) != hipSuccess) return (int)e;
    } else {
        if ((e = hipMemsetAsync(gx, 0, nx * sizeof(float), s)) != hipSuccess) return (int)e;
        if ((e = hipMemsetAsync(gy, 0, ny * sizeof(float), s)) != hipSuccess) return (int)e;
    }
    dim3 grid(cdiv(m > n ? m : n, LP_THREADS), b, 2);
    hipLaunchKernelGGL(chamfer_gram_grad_kernel, grid, dim3(LP_THREADS), 0, s, m, n, d, x, y, gminx, argx, gminy, argy, gx, gy,
                       guni, gscale);
    return pdgn_launch_status();
}

extern "C" int pdgn_chamfer_gram_grad(int b, int m, int n, int d, const float *x, const float *y,
                                      const float *gminx, const int32_t *argx, const float *gminy,
                                      const int32_t *argy, float *gx, float *gy, pdgn_stream_t stream) {
    if (b < 0 || m < 1 || n < 1 || d < 1 || d > CH_MAXD || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    return chamfer_grad_launch(b, m, n, d, x, y, gminx, argx, gminy, argy, gx, gy, nullptr, 0.f, (hipStream_t)stream);
}

// The same with a uniform upstream gradient g[0] * scale for every minimum (the loss is scale * sum of the minima,
// utils/chamfer_loss.py:16-20): no expanded gradient tensor.
extern "C" int pdgn_chamfer_gram_grad_uniform(int b, int m, int n, int d, const float *x, const float *y, const float *g,
                                              float scale, const int32_t *argx, const int32_t *argy, float *gx, float *gy,
                                              pdgn_stream_t stream) {
    if (b < 0 || m < 1 || n < 1 || d < 1 || d > CH_MAXD || b > 65535 || !g) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    return chamfer_grad_launch(b, m, n, d, x, y, nullptr, argx, nullptr, argy, gx, gy, g, scale, (hipStream_t)stream);
}
