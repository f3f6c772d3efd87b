// bnact.hip -- fused BatchNorm + activation over channels-last (rows x C) activations.
//
// Every conv of the reference's deconvolution block is followed by BatchNorm2d (training mode:
// batch statistics over (B,N,k)) and LeakyReLU/ReLU (models/PDGNet_v2.py:561-565, 603-625, 537-545),
// each a separate pass over a (B,C,N,k) tensor.  In the point-major layout used here these are
// (rows x C) matrices with C contiguous, and the whole chain is two streaming kernels forward
// (statistics; normalise+activate) and two backward (the two batch reductions; the input gradient):
//   forward : y  = act(x * scale[c] + shift[c]),  scale = gamma*invstd, shift = beta - mean*scale
//   backward: dz = dy * act'(z);  s1 = sum dz;  s2 = sum dz * xhat
//             dx = scale * (dz - s1/R - xhat * s2/R)      (training)   |   dx = scale * dz  (eval)
// All four are HBM-bound: float4 per lane, a wave covers 1 KiB of one row (or several short rows).
// Per-block partial sums are fp32 over <= 64K rows; they are combined in fp64 by tiny finalize
// kernels (two-stage reduction: no atomics, bitwise reproducible).
#include "common.h"

#include "bn_geom.h"
#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_LEAKY 2   // negative_slope 0.01 (nn.LeakyReLU default used throughout the reference)

__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act == ACT_RELU) return fmaxf(z, 0.f);
    if (act == ACT_LEAKY) return z > 0.f ? z : 0.01f * z;
    return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
    if (act == ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == ACT_LEAKY) return z > 0.f ? 1.f : 0.01f;
    return 1.f;
}

// Geometry shared by the reduction kernels: CG = C/4 float4 column groups; thread t owns column
// group t % CGB of the block's column slice and row lane t / CGB.
struct ClGeom {
    int cg, cgb, rl;     // column groups total, per block (<= 256, divides 256 or equals cg), row lanes
};

// part[blockIdx.y][c] = sum_r x[r,c];  part[blockIdx.y][C + c] = sum_r x[r,c]^2 over the block's rows
// (fp32 partials over <= 64K rows; the finalize kernels combine them in fp64 -- no atomics).
// The sums are taken of x - pivot, pivot = row 0 of x (the same for every block): E[x^2] - mean^2 on raw sums cancels
// quadratically in |mean| / std, on shifted sums it does not (the finalize kernel adds the pivot back).
__global__ __launch_bounds__(BN_THREADS) void cl_stats_kernel(long long R, int C, int cgb, int rows_per_block,
                                                              const float *__restrict__ x,
                                                              float *__restrict__ part) {
    __shared__ float4 red[2][BN_THREADS];
    const int cgl = threadIdx.x % cgb, rlane = threadIdx.x / cgb, rl = BN_THREADS / cgb;
    const int cgi = blockIdx.x * cgb + cgl;                 // global column group
    const bool cok = cgi * 4 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(R, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok) {
        const float *X = x + cgi * 4;
        const float4 pv = *reinterpret_cast<const float4 *>(X);                 // pivot: row 0
        long long r = r0 + rlane;
        for (; r + 3LL * rl < r1; r += 4LL * rl) {              // four independent row loads in flight
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(X + (r + (long long)u * rl) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float dx = v[u].x - pv.x, dy = v[u].y - pv.y, dz = v[u].z - pv.z, dw = v[u].w - pv.w;
                s.x += dx; s.y += dy; s.z += dz; s.w += dw;
                q.x = __fmaf_rn(dx, dx, q.x); q.y = __fmaf_rn(dy, dy, q.y);
                q.z = __fmaf_rn(dz, dz, q.z); q.w = __fmaf_rn(dw, dw, q.w);
            }
        }
        for (; r < r1; r += rl) {
            const float4 v = *reinterpret_cast<const float4 *>(X + r * C);
            const float dx = v.x - pv.x, dy = v.y - pv.y, dz = v.z - pv.z, dw = v.w - pv.w;
            s.x += dx; s.y += dy; s.z += dz; s.w += dw;
            q.x = __fmaf_rn(dx, dx, q.x); q.y = __fmaf_rn(dy, dy, q.y);
            q.z = __fmaf_rn(dz, dz, q.z); q.w = __fmaf_rn(dw, dw, q.w);
        }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (rlane == 0 && cok) {
        for (int j = 1; j < rl; ++j) {
            float4 a = red[0][j * cgb + cgl], b = red[1][j * cgb + cgl];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
        }
        float *P = part + (size_t)blockIdx.y * 2 * C;
        *reinterpret_cast<float4 *>(P + cgi * 4) = s;
        *reinterpret_cast<float4 *>(P + C + cgi * 4) = q;
    }
}

// From the fp64 sums: mean, biased var -> scale/shift/mean/invstd; running stats with momentum and
// the unbiased variance (nn.BatchNorm semantics).  stats out: [scale | shift | mean | invstd] (4C).
#define FIN_CH 4       // channels per finalize block
#define FIN_PL 64      // part lanes per channel (short dependent-load chains: nparts/64 steps)

// Sum `nparts` fp32 partials of channel-slot `c` (stride `ld`) in fp64: 64 lanes per channel, LDS tree.
__device__ __forceinline__ double fin_reduce(const float *__restrict__ part, int nparts, size_t ld, int c, bool ok,
                                             double (*red)[FIN_CH]) {
    const int cl = threadIdx.x % FIN_CH, pl = threadIdx.x / FIN_CH;
    double s = 0;
    if (ok) {
        int p = pl;
        for (; p + 3 * FIN_PL < nparts; p += 4 * FIN_PL) {          // four independent loads in flight
            const float a0 = part[(size_t)p * ld + c], a1 = part[(size_t)(p + FIN_PL) * ld + c];
            const float a2 = part[(size_t)(p + 2 * FIN_PL) * ld + c], a3 = part[(size_t)(p + 3 * FIN_PL) * ld + c];
            s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        }
        for (; p < nparts; p += FIN_PL) s += (double)part[(size_t)p * ld + c];
    }
    red[pl][cl] = s;
    __syncthreads();
    for (int h = FIN_PL / 2; h > 0; h >>= 1) {
        if (pl < h) red[pl][cl] += red[pl + h][cl];
        __syncthreads();
    }
    return red[0][cl];
}

// Both column sums of a channel (slots c and C + c) in ONE pass: the loads of the two reductions are in flight together
// and the LDS tree carries both values -- these finalize kernels are ~150 launches per step, each a pure latency chain.
__device__ __forceinline__ void fin_reduce2(const float *__restrict__ part, int nparts, size_t ld, int c, int C, bool ok,
                                            double (*red)[FIN_CH], double (*red2)[FIN_CH], double &s1, double &s2) {
    const int cl = threadIdx.x % FIN_CH, pl = threadIdx.x / FIN_CH;
    double a = 0, b = 0;
    if (ok) {
        int p = pl;
        for (; p + 3 * FIN_PL < nparts; p += 4 * FIN_PL) {          // eight independent loads in flight
            const float *q = part + (size_t)p * ld + c;
            const float a0 = q[0], a1 = q[(size_t)FIN_PL * ld], a2 = q[(size_t)2 * FIN_PL * ld], a3 = q[(size_t)3 * FIN_PL * ld];
            const float b0 = q[C], b1 = q[(size_t)FIN_PL * ld + C], b2 = q[(size_t)2 * FIN_PL * ld + C], b3 = q[(size_t)3 * FIN_PL * ld + C];
            a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
        }
        for (; p < nparts; p += FIN_PL) {
            a += (double)part[(size_t)p * ld + c];
            b += (double)part[(size_t)p * ld + C + c];
        }
    }
    red[pl][cl] = a;
    red2[pl][cl] = b;
    __syncthreads();
    for (int h = FIN_PL / 2; h > 0; h >>= 1) {
        if (pl < h) {
            red[pl][cl] += red[pl + h][cl];
            red2[pl][cl] += red2[pl + h][cl];
        }
        __syncthreads();
    }
    s1 = red[0][cl];
    s2 = red2[0][cl];
}

__global__ __launch_bounds__(FIN_CH * FIN_PL) void cl_finalize_kernel(
    long long R, int C, int nparts, float eps, float momentum, const float *__restrict__ part,
    const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ pre_bias,
    float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ stats,
    const float *__restrict__ pivot) {                      // the sums are of (x - pivot[c]) when given (cl_stats_kernel)
    __shared__ double red[FIN_PL][FIN_CH], red2[FIN_PL][FIN_CH];
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH;
    const bool ok = c < C;
    double s1, s2;
    fin_reduce2(part, nparts, (size_t)2 * C, c, C, ok, red, red2, s1, s2);
    if (!ok || threadIdx.x >= FIN_CH) return;
    const double ms = s1 / (double)R;                       // mean of the (shifted) values
    double var = s2 / (double)R - ms * ms;
    var = var < 0 ? 0 : var;
    const double mean = ms + (pivot ? (double)pivot[c] : 0.0);
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float scale = g * invstd;
    stats[c] = scale;
    stats[C + c] = b - (float)mean * scale;
    stats[2 * C + c] = (float)mean;
    stats[3 * C + c] = invstd;
    if (running_mean) {
        const double unbiased = R > 1 ? var * (double)R / (double)(R - 1) : var;
        // pre_bias: the producer's bias, left out of x (BatchNorm(x + b) = BatchNorm(x)); only the running mean sees it
        const float mb = (float)mean + (pre_bias ? pre_bias[c] : 0.f);
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mb;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// The same finalisation from BLOCK-SHIFTED partials (the epilogues of pdgn_gemm_nt / pdgn_gemm_nn / pdgn_thin_nt): partial
// row b covers rows b*rpp .. of x and holds [sum (x - pv_b) | sum (x - pv_b)^2 | pv_b] with pv_b the block's own first row.
// Per block: mean_b = pv_b + s/n, M2_b = q - s^2/n (no cancellation: |x - pv_b| ~ std); the blocks are combined in fp64:
// mean = sum n_b mean_b / R, var = (sum M2_b + sum n_b mean_b^2) / R - mean^2.
__global__ __launch_bounds__(FIN_CH * FIN_PL) void cl_finalize_blocks_kernel(
    long long R, int C, int nparts, int rpp, float eps, float momentum, const float *__restrict__ part,
    const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ pre_bias,
    float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ stats) {
    __shared__ double red[FIN_PL][FIN_CH], red2[FIN_PL][FIN_CH];
    const int cl = threadIdx.x % FIN_CH, pl = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + cl;
    const bool ok = c < C;
    double a = 0, b = 0;
    const double inv_full = 1.0 / (double)rpp;                     // every block but the last is full: no fp64 division per block
    const int pmax = (int)min((long long)nparts, (R + rpp - 1) / rpp);      // blocks that hold rows
    if (ok)
#pragma unroll 4
        for (int p = pl; p < pmax; p += FIN_PL) {                 // (no exit inside: the loads of four blocks go out together)
            const long long left = R - (long long)p * rpp;
            const bool full = left >= rpp;
            const double n = (double)(full ? rpp : left), inv = full ? inv_full : 1.0 / (double)left;
            const float *q = part + (size_t)p * 3 * C + c;
            const double s = (double)q[0], sq = (double)q[C], pv = (double)q[2 * C];
            const double mb = pv + s * inv;
            a += n * mb;
            b += (sq - s * s * inv) + n * mb * mb;
        }
    red[pl][cl] = a;
    red2[pl][cl] = b;
    __syncthreads();
    for (int h = FIN_PL / 2; h > 0; h >>= 1) {
        if (pl < h) {
            red[pl][cl] += red[pl + h][cl];
            red2[pl][cl] += red2[pl + h][cl];
        }
        __syncthreads();
    }
    if (!ok || threadIdx.x >= FIN_CH) return;
    const double mean = red[0][cl] / (double)R;
    double var = red2[0][cl] / (double)R - mean * mean;
    var = var < 0 ? 0 : var;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const float scale = g * invstd;
    stats[c] = scale;
    stats[C + c] = be - (float)mean * scale;
    stats[2 * C + c] = (float)mean;
    stats[3 * C + c] = invstd;
    if (running_mean) {
        const double unbiased = R > 1 ? var * (double)R / (double)(R - 1) : var;
        const float mb = (float)mean + (pre_bias ? pre_bias[c] : 0.f);
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mb;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// The same in two launches for long partial lists (a 358400-row GEMM leaves 2800-5600 partial rows; one workgroup per four
// channels walking all of them took 35-60 us on the issuing stream, twice per block and pass): slice s of FB_SLICES(nparts)
// sums its share of the blocks in fp64 into scratch[s][2][C]; the second kernel adds the slices in order (deterministic)
// and finalises.
__global__ __launch_bounds__(FIN_CH * FIN_PL) void cl_finalize_blocks_slice_kernel(
    long long R, int C, int nparts, int rpp, int per_slice, const float *__restrict__ part, double *__restrict__ scratch) {
    __shared__ double red[FIN_PL][FIN_CH], red2[FIN_PL][FIN_CH];
    const int cl = threadIdx.x % FIN_CH, pl = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + cl;
    const bool ok = c < C;
    double a = 0, b = 0;
    const double inv_full = 1.0 / (double)rpp;
    const int pmax_all = (int)min((long long)nparts, (R + rpp - 1) / rpp);
    const int p0 = blockIdx.y * per_slice, pmax = min(pmax_all, p0 + per_slice);
    if (ok)
#pragma unroll 4
        for (int p = p0 + pl; p < pmax; p += FIN_PL) {
            const long long left = R - (long long)p * rpp;
            const bool full = left >= rpp;
            const double n = (double)(full ? rpp : left), inv = full ? inv_full : 1.0 / (double)left;
            const float *q = part + (size_t)p * 3 * C + c;
            const double s = (double)q[0], sq = (double)q[C], pv = (double)q[2 * C];
            const double mb = pv + s * inv;
            a += n * mb;
            b += (sq - s * s * inv) + n * mb * mb;
        }
    red[pl][cl] = a;
    red2[pl][cl] = b;
    __syncthreads();
    for (int h = FIN_PL / 2; h > 0; h >>= 1) {
        if (pl < h) {
            red[pl][cl] += red[pl + h][cl];
            red2[pl][cl] += red2[pl + h][cl];
        }
        __syncthreads();
    }
    if (!ok || threadIdx.x >= FIN_CH) return;
    scratch[((size_t)blockIdx.y * 2) * C + c] = red[0][cl];
    scratch[((size_t)blockIdx.y * 2 + 1) * C + c] = red2[0][cl];
}

// The same for 64-channel multiples (round 6): a workgroup takes 64 CONSECUTIVE channels (a wave reads 256 contiguous bytes of a
// partial row: the four-channel form above touches 16 B of every row per workgroup) on four part lanes; more, shorter slices.  The
// row-panel kernel's partials (gemm_rp.hip: one row per 32 rows of the result, 11200 x [3 x 512] for the stage-4 edge-level layer) took
// 100 us in the four-channel form.
__global__ __launch_bounds__(256) void cl_finalize_blocks_slice64_kernel(
    long long R, int C, int nparts, int rpp, int per_slice, const float *__restrict__ part, double *__restrict__ scratch) {
    __shared__ double red[4][64], red2[4][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double a = 0, b = 0;
    const double inv_full = 1.0 / (double)rpp;
    const int pmax_all = (int)min((long long)nparts, (R + rpp - 1) / rpp);
    const int p0 = blockIdx.y * per_slice, pmax = min(pmax_all, p0 + per_slice);
#pragma unroll 4
    for (int p = p0 + pl; p < pmax; p += 4) {
        const long long left = R - (long long)p * rpp;
        const bool full = left >= rpp;
        const double n = (double)(full ? rpp : left), inv = full ? inv_full : 1.0 / (double)left;
        const float *q = part + (size_t)p * 3 * C + c;
        const double s = (double)q[0], sq = (double)q[C], pv = (double)q[2 * C];
        const double mb = pv + s * inv;
        a += n * mb;
        b += (sq - s * s * inv) + n * mb * mb;
    }
    red[pl][cl] = a;
    red2[pl][cl] = b;
    __syncthreads();
    if (pl == 0) {                                                // (in lane order: a fixed order of summation)
        scratch[((size_t)blockIdx.y * 2) * C + c] = ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
        scratch[((size_t)blockIdx.y * 2 + 1) * C + c] = ((red2[0][cl] + red2[1][cl]) + red2[2][cl]) + red2[3][cl];
    }
}

__global__ void cl_finalize_blocks_join_kernel(long long R, int C, int slices, float eps, float momentum,
                                               const double *__restrict__ scratch, const float *__restrict__ gamma,
                                               const float *__restrict__ beta, const float *__restrict__ pre_bias,
                                               float *__restrict__ running_mean, float *__restrict__ running_var,
                                               float *__restrict__ stats) {
    // 64 channels x 4 lanes per workgroup (blockDim = 256): lane l adds slices l, l + 4, ... in order, the four sums meet in lane
    // order -- a fixed order of summation; a single lane walking 128 slices was 128 dependent loads
    __shared__ double ra[4][64], rb[4][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double a = 0, b = 0;
    if (c < C)
        for (int s = pl; s < slices; s += 4) {
            a += scratch[((size_t)s * 2) * C + c];
            b += scratch[((size_t)s * 2 + 1) * C + c];
        }
    ra[pl][cl] = a;
    rb[pl][cl] = b;
    __syncthreads();
    if (pl != 0 || c >= C) return;
    a = ((ra[0][cl] + ra[1][cl]) + ra[2][cl]) + ra[3][cl];
    b = ((rb[0][cl] + rb[1][cl]) + rb[2][cl]) + rb[3][cl];
    const double mean = a / (double)R;
    double var = b / (double)R - mean * mean;
    var = var < 0 ? 0 : var;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const float scale = g * invstd;
    stats[c] = scale;
    stats[C + c] = be - (float)mean * scale;
    stats[2 * C + c] = (float)mean;
    stats[3 * C + c] = invstd;
    if (running_mean) {
        const double unbiased = R > 1 ? var * (double)R / (double)(R - 1) : var;
        const float mb = (float)mean + (pre_bias ? pre_bias[c] : 0.f);
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mb;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

static inline int fb_slices(int c, long long nparts) {           // 0: the one-launch form
    // enough workgroups (c / 4 channel groups x slices ~ 512) with at least two passes of the 64 part lanes each; measured
    // (tools/finalize_bench.py): 5600 x 64 channels 35 -> 14 us with 32 slices; 2800 x 512 is no faster sliced 32 ways (25 -> 30)
    if (nparts < 512) return 0;
    if (c % 64 == 0) {                                              // the 64-channel form: ~16 partial rows per part lane, at most 128 slices
        // (tools/finalize_bench.py, round 6: 11200 x 512: 87 -> 36 us, 5600 x 256: 44 -> 17, 2800 x 512: 25 -> 12, 5600 x 64: 33 -> 16;
        //  below ~1400 rows one launch is as fast: 560 x 512 5.5 vs 10 us)
        if (nparts <= 1400) return 0;                               // (1400: a 358400-row layer on the row-panel kernel, one partial row per 256-row panel)
        long long s64 = nparts / 64;
        return (int)(s64 < 2 ? 2 : (s64 > 128 ? 128 : s64));
    }
    long long s = 512 / (c / 4 > 0 ? c / 4 : 1);
    s = s > 32 ? 32 : s;
    while (s > 1 && nparts / s < 128) s >>= 1;
    return s < 2 ? 0 : (int)s;
}

// eval mode: scale/shift from the running statistics
__global__ void cl_eval_stats_kernel(int C, float eps, const float *__restrict__ gamma, const float *__restrict__ beta,
                                     const float *__restrict__ pre_bias, const float *__restrict__ running_mean,
                                     const float *__restrict__ running_var, float *__restrict__ stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = rsqrtf(running_var[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float m = running_mean[c] - (pre_bias ? pre_bias[c] : 0.f);     // mean of x when x excludes the producer's bias
    stats[c] = g * invstd;
    stats[C + c] = b - m * g * invstd;
    stats[2 * C + c] = m;
    stats[3 * C + c] = invstd;
}

// y = act(x*scale + shift) [* mul].  Same geometry as the reductions: a thread owns one float4
// column group (its scale/shift live in registers) and streams its rows four at a time.
// pn > 0 (INTERLEAVED y / dy): x row b*pn + n, channel 2c + j  <->  y row b*2pn + j*pn + n, channel c -- the reference's
// (B,2Fout,N,1) -> view(B,Fout,2,N) -> (B,Fout,2N) of a deconvolution block's result (models/PDGNet_v2.py:645-647) in
// point-major form, written / read here instead of by a separate permute copy.
__device__ __forceinline__ void cl_interleaved_rows(long long r, int pn, long long &ra, long long &rb) {
    const long long b = r / pn;
    ra = b * 2 * pn + (r - b * pn);
    rb = ra + pn;
}

__global__ __launch_bounds__(BN_THREADS) void cl_apply_kernel(long long R, int C, int cgb, int rows_per_block, int act,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ stats,
                                                              const float *__restrict__ mul, float *__restrict__ y, int pn) {
    const int cgl = threadIdx.x % cgb, rlane = threadIdx.x / cgb, rl = BN_THREADS / cgb;
    const int cgi = blockIdx.x * cgb + cgl;
    if (cgi * 4 >= C) return;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(R, r0 + rows_per_block);
    const float4 sc = *reinterpret_cast<const float4 *>(stats + cgi * 4);
    const float4 sh = *reinterpret_cast<const float4 *>(stats + C + cgi * 4);
    const bool has_mul = mul != nullptr;
    for (long long r = r0 + rlane; r < r1; r += 4 * rl) {
        float4 v[4], m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long rr = r + (long long)u * rl;
            if (rr < r1) {
                v[u] = *reinterpret_cast<const float4 *>(x + rr * C + cgi * 4);
                if (has_mul) m[u] = *reinterpret_cast<const float4 *>(mul + rr * C + cgi * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long rr = r + (long long)u * rl;
            if (rr < r1) {
                float4 o;
                o.x = act_fwd(__fmaf_rn(v[u].x, sc.x, sh.x), act); o.y = act_fwd(__fmaf_rn(v[u].y, sc.y, sh.y), act);
                o.z = act_fwd(__fmaf_rn(v[u].z, sc.z, sh.z), act); o.w = act_fwd(__fmaf_rn(v[u].w, sc.w, sh.w), act);
                if (has_mul) { o.x *= m[u].x; o.y *= m[u].y; o.z *= m[u].z; o.w *= m[u].w; }
                if (pn) {
                    long long ra, rb;
                    cl_interleaved_rows(rr, pn, ra, rb);
                    *reinterpret_cast<float2 *>(y + ra * (C / 2) + cgi * 2) = make_float2(o.x, o.z);
                    *reinterpret_cast<float2 *>(y + rb * (C / 2) + cgi * 2) = make_float2(o.y, o.w);
                } else {
                    *reinterpret_cast<float4 *>(y + rr * C + cgi * 4) = o;
                }
            }
        }
    }
}

// bsums[c] += sum dz ; bsums[C+c] += sum dz*xhat   (dz = dy [* mul] * act'(z))
__global__ __launch_bounds__(BN_THREADS) void cl_bwd_reduce_kernel(long long R, int C, int cgb, int rows_per_block,
                                                                   int act, const float *__restrict__ x,
                                                                   const float *__restrict__ dy,
                                                                   const float *__restrict__ mul,
                                                                   const float *__restrict__ stats,
                                                                   float *__restrict__ part, int pn) {
    __shared__ float4 red[2][BN_THREADS];
    const int cgl = threadIdx.x % cgb, rlane = threadIdx.x / cgb, rl = BN_THREADS / cgb;
    const int cgi = blockIdx.x * cgb + cgl;
    const bool cok = cgi * 4 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(R, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok) {
        const float4 sc = *reinterpret_cast<const float4 *>(stats + cgi * 4);
        const float4 sh = *reinterpret_cast<const float4 *>(stats + C + cgi * 4);
        const float4 mu = *reinterpret_cast<const float4 *>(stats + 2 * C + cgi * 4);
        const float4 is = *reinterpret_cast<const float4 *>(stats + 3 * C + cgi * 4);
        const bool has_mul = mul != nullptr;
        for (long long rb = r0 + rlane; rb < r1; rb += 4LL * rl) {  // four rows (8-12 independent loads) in flight
            float4 v[4], g[4], m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long r = rb + (long long)u * rl;
                if (r < r1) {
                    v[u] = *reinterpret_cast<const float4 *>(x + r * C + cgi * 4);
                    if (pn) {
                        long long ra, rbb;
                        cl_interleaved_rows(r, pn, ra, rbb);
                        const float2 ga = *reinterpret_cast<const float2 *>(dy + ra * (C / 2) + cgi * 2);
                        const float2 gb = *reinterpret_cast<const float2 *>(dy + rbb * (C / 2) + cgi * 2);
                        g[u] = make_float4(ga.x, gb.x, ga.y, gb.y);
                    } else {
                        g[u] = *reinterpret_cast<const float4 *>(dy + r * C + cgi * 4);
                    }
                    if (has_mul) m[u] = *reinterpret_cast<const float4 *>(mul + r * C + cgi * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (rb + (long long)u * rl < r1) {
                    if (has_mul) { g[u].x *= m[u].x; g[u].y *= m[u].y; g[u].z *= m[u].z; g[u].w *= m[u].w; }
                    float dz;
                    dz = g[u].x * act_grad(__fmaf_rn(v[u].x, sc.x, sh.x), act); s.x += dz; q.x = __fmaf_rn(dz, (v[u].x - mu.x) * is.x, q.x);
                    dz = g[u].y * act_grad(__fmaf_rn(v[u].y, sc.y, sh.y), act); s.y += dz; q.y = __fmaf_rn(dz, (v[u].y - mu.y) * is.y, q.y);
                    dz = g[u].z * act_grad(__fmaf_rn(v[u].z, sc.z, sh.z), act); s.z += dz; q.z = __fmaf_rn(dz, (v[u].z - mu.z) * is.z, q.z);
                    dz = g[u].w * act_grad(__fmaf_rn(v[u].w, sc.w, sh.w), act); s.w += dz; q.w = __fmaf_rn(dz, (v[u].w - mu.w) * is.w, q.w);
                }
            }
        }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (rlane == 0 && cok) {
        for (int j = 1; j < rl; ++j) {
            float4 a = red[0][j * cgb + cgl], b = red[1][j * cgb + cgl];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
        }
        float *P = part + (size_t)blockIdx.y * 2 * C;
        *reinterpret_cast<float4 *>(P + cgi * 4) = s;
        *reinterpret_cast<float4 *>(P + C + cgi * 4) = q;
    }
}

// bsums[c] = sum over partials (fp64 accumulation, fp32 result): [sum dz | sum dz*xhat], and the
// per-channel coefficients of the input gradient: dx = scale*dz - ca - cb*x with
//   cb = scale*invstd*mean(dz*xhat),  ca = scale*mean(dz) - cb*mean   (zero in eval mode).
__global__ __launch_bounds__(FIN_CH * FIN_PL) void cl_bwd_finalize_kernel(long long R, int C, int nparts, int training,
                                                                         const float *__restrict__ part,
                                                                         const float *__restrict__ stats,
                                                                         float *__restrict__ bsums,
                                                                         float *__restrict__ coef) {
    __shared__ double red[FIN_PL][FIN_CH], red2[FIN_PL][FIN_CH];
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH;
    const bool ok = c < C;
    double s1, s2;
    fin_reduce2(part, nparts, (size_t)2 * C, c, C, ok, red, red2, s1, s2);
    if (!ok || threadIdx.x >= FIN_CH) return;
    bsums[c] = (float)s1;
    bsums[C + c] = (float)s2;
    float ca = 0.f, cb = 0.f;
    if (training) {
        const float sc = stats[c], mu = stats[2 * C + c], is = stats[3 * C + c];
        cb = sc * is * (float)(s2 / (double)R);
        ca = sc * (float)(s1 / (double)R) - cb * mu;
    }
    coef[c] = ca;
    coef[C + c] = cb;
}

// dx = scale*dz - ca - cb*x   (dz = dy [* mul] * act'(z));  optionally dmul = dy*act(z)
__global__ __launch_bounds__(BN_THREADS) void cl_bwd_apply_kernel(long long R, int C, int cgb, int rows_per_block, int act,
                                                                  const float *__restrict__ x,
                                                                  const float *__restrict__ dy,
                                                                  const float *__restrict__ mul,
                                                                  const float *__restrict__ stats,
                                                                  const float *__restrict__ coef,
                                                                  float *__restrict__ dx, float *__restrict__ dmul, int pn) {
    const int cgl = threadIdx.x % cgb, rlane = threadIdx.x / cgb, rl = BN_THREADS / cgb;
    const int cgi = blockIdx.x * cgb + cgl;
    if (cgi * 4 >= C) return;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(R, r0 + rows_per_block);
    float sc[4], sh[4], ca[4], cb[4];
    *reinterpret_cast<float4 *>(sc) = *reinterpret_cast<const float4 *>(stats + cgi * 4);
    *reinterpret_cast<float4 *>(sh) = *reinterpret_cast<const float4 *>(stats + C + cgi * 4);
    *reinterpret_cast<float4 *>(ca) = *reinterpret_cast<const float4 *>(coef + cgi * 4);
    *reinterpret_cast<float4 *>(cb) = *reinterpret_cast<const float4 *>(coef + C + cgi * 4);
    const bool has_mul = mul != nullptr, want_dmul = dmul != nullptr;
    for (long long r = r0 + rlane; r < r1; r += 4 * rl) {
        float v[4][4], g[4][4], m[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long rr = r + (long long)u * rl;
            if (rr < r1) {
                *reinterpret_cast<float4 *>(v[u]) = *reinterpret_cast<const float4 *>(x + rr * C + cgi * 4);
                if (pn) {
                    long long ra, rb;
                    cl_interleaved_rows(rr, pn, ra, rb);
                    const float2 ga = *reinterpret_cast<const float2 *>(dy + ra * (C / 2) + cgi * 2);
                    const float2 gb = *reinterpret_cast<const float2 *>(dy + rb * (C / 2) + cgi * 2);
                    g[u][0] = ga.x; g[u][1] = gb.x; g[u][2] = ga.y; g[u][3] = gb.y;
                } else {
                    *reinterpret_cast<float4 *>(g[u]) = *reinterpret_cast<const float4 *>(dy + rr * C + cgi * 4);
                }
                if (has_mul) *reinterpret_cast<float4 *>(m[u]) = *reinterpret_cast<const float4 *>(mul + rr * C + cgi * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long rr = r + (long long)u * rl;
            if (rr < r1) {
                float o[4], om[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float z = __fmaf_rn(v[u][j], sc[j], sh[j]);
                    om[j] = g[u][j] * act_fwd(z, act);
                    const float dz = (has_mul ? g[u][j] * m[u][j] : g[u][j]) * act_grad(z, act);
                    o[j] = __fmaf_rn(sc[j], dz, -ca[j]) - cb[j] * v[u][j];
                }
                *reinterpret_cast<float4 *>(dx + rr * C + cgi * 4) = *reinterpret_cast<float4 *>(o);
                if (want_dmul) *reinterpret_cast<float4 *>(dmul + rr * C + cgi * 4) = *reinterpret_cast<float4 *>(om);
            }
        }
    }
}

// ---------------------------------------------------------------------------- C ABI
// Floats of scratch the two reductions need for (rows, c): one [2c] partial per row-block.
extern "C" long long pdgn_bn_scratch_floats(long long rows, int c) {
    if (rows < 1 || c < 4 || c % 4) return PDGN_ERR_INVALID;
    int cgb, gx, gy, rpb;
    cl_geometry(rows, c, &cgb, &gx, &gy, &rpb);
    return (long long)gy * 2 * c + 2 * c;                  // partials + the backward's coefficient row
}

extern "C" int pdgn_bn_stats(long long rows, int c, float eps, float momentum, const float *x, const float *gamma,
                             const float *beta, const float *pre_bias, float *running_mean, float *running_var,
                             float *scratch, float *stats, pdgn_stream_t stream) {
    if (rows < 1 || c < 4 || c % 4) return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int cgb, gx, gy, rpb;
    cl_geometry(rows, c, &cgb, &gx, &gy, &rpb);
    hipLaunchKernelGGL(cl_stats_kernel, dim3(gx, gy), dim3(BN_THREADS), 0, s, rows, c, cgb, rpb, x, scratch);
    hipLaunchKernelGGL(cl_finalize_kernel, dim3(cdiv(c, FIN_CH)), dim3(FIN_CH * FIN_PL), 0, s, rows, c, gy, eps, momentum, scratch,
                       gamma, beta, pre_bias, running_mean, running_var, stats, x);
    return pdgn_launch_status();
}

// Second stage only: `scratch` already holds the per-row-block partial sums (written by a producer's epilogue,
// e.g. pdgn_window_gather_sum_stats, in the geometry of bn_geom.h).
extern "C" int pdgn_bn_stats_from_partials(long long rows, int c, float eps, float momentum, const float *gamma,
                                           const float *beta, const float *pre_bias, float *running_mean,
                                           float *running_var, const float *scratch, float *stats, pdgn_stream_t stream) {
    if (rows < 1 || c < 4 || c % 4) return PDGN_ERR_INVALID;
    int cgb, gx, gy, rpb;
    cl_geometry(rows, c, &cgb, &gx, &gy, &rpb);
    hipLaunchKernelGGL(cl_finalize_kernel, dim3(cdiv(c, FIN_CH)), dim3(FIN_CH * FIN_PL), 0, (hipStream_t)stream, rows, c, gy, eps,
                       momentum, scratch, gamma, beta, pre_bias, running_mean, running_var, stats, (const float *)nullptr);
    return pdgn_launch_status();
}

// The second stage over the BLOCK-SHIFTED partials a GEMM epilogue writes: `nparts` rows of [3c] floats (per-column
// sum (x - pv) | sum (x - pv)^2 | pv of a block of `block_rows` rows; pdgn_gemm_nt_stat_rows / _stat_block_rows,
// pdgn_thin_stat_rows / _stat_block_rows).
extern "C" int pdgn_bn_stats_from_gemm_partials(long long rows, int c, long long nparts, int block_rows, float eps,
                                                float momentum, const float *gamma, const float *beta, const float *pre_bias,
                                                float *running_mean, float *running_var, const float *partials,
                                                float *stats, double *scratch, pdgn_stream_t stream) {
    if (rows < 1 || c < 4 || c % 4 || nparts < 1 || nparts > 0x7fffffffLL || block_rows < 1 ||
        nparts * (long long)block_rows < rows)
        return PDGN_ERR_INVALID;
    const int slices = scratch ? fb_slices(c, nparts) : 0;
    if (slices > 1) {
        const int per = (int)((nparts + slices - 1) / slices);
        if (c % 64 == 0)
            hipLaunchKernelGGL(cl_finalize_blocks_slice64_kernel, dim3(c / 64, slices), dim3(256), 0, (hipStream_t)stream, rows, c, (int)nparts,
                               block_rows, per, partials, scratch);
        else
            hipLaunchKernelGGL(cl_finalize_blocks_slice_kernel, dim3(cdiv(c, FIN_CH), slices), dim3(FIN_CH * FIN_PL), 0,
                               (hipStream_t)stream, rows, c, (int)nparts, block_rows, per, partials, scratch);
        hipLaunchKernelGGL(cl_finalize_blocks_join_kernel, dim3(cdiv(c, 64)), dim3(256), 0, (hipStream_t)stream, rows, c, slices,
                           eps, momentum, (const double *)scratch, gamma, beta, pre_bias, running_mean, running_var, stats);
        return pdgn_launch_status();
    }
    hipLaunchKernelGGL(cl_finalize_blocks_kernel, dim3(cdiv(c, FIN_CH)), dim3(FIN_CH * FIN_PL), 0, (hipStream_t)stream, rows, c,
                       (int)nparts, block_rows, eps, momentum, partials, gamma, beta, pre_bias, running_mean, running_var, stats);
    return pdgn_launch_status();
}

// fp64 scratch elements pdgn_bn_stats_from_gemm_partials wants for `nparts` partial rows of c channels (0: none needed).
extern "C" long long pdgn_bn_blocks_scratch_doubles(int c, long long nparts) {
    if (c < 1 || nparts < 1) return PDGN_ERR_INVALID;
    return (long long)fb_slices(c, nparts) * 2 * c;
}

extern "C" int pdgn_bn_eval_stats(int c, float eps, const float *gamma, const float *beta, const float *pre_bias,
                                  const float *running_mean, const float *running_var, float *stats,
                                  pdgn_stream_t stream) {
    if (c < 1) return PDGN_ERR_INVALID;
    hipLaunchKernelGGL(cl_eval_stats_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, c, eps, gamma,
                       beta, pre_bias, running_mean, running_var, stats);
    return pdgn_launch_status();
}

extern "C" int pdgn_bn_act_forward(long long rows, int c, int act, const float *x, const float *stats, const float *mul,
                                   float *y, int interleave_n, pdgn_stream_t stream) {
    if (rows < 1 || c < 4 || c % 4 || act < 0 || act > 2) return PDGN_ERR_INVALID;
    if (interleave_n < 0 || (interleave_n > 0 && (rows % interleave_n || mul))) return PDGN_ERR_INVALID;
    int cgb, gx, gy, rpb;
    cl_geometry(rows, c, &cgb, &gx, &gy, &rpb);
    hipLaunchKernelGGL(cl_apply_kernel, dim3(gx, gy), dim3(BN_THREADS), 0, (hipStream_t)stream, rows, c, cgb, rpb, act,
                       x, stats, mul, y, interleave_n);
    return pdgn_launch_status();
}

extern "C" int pdgn_bn_act_backward(long long rows, int c, int act, int training, const float *x, const float *dy,
                                    const float *mul, const float *stats, float *scratch, float *bsums, float *dx,
                                    float *dmul, int interleave_n, pdgn_stream_t stream) {
    if (rows < 1 || c < 4 || c % 4 || act < 0 || act > 2) return PDGN_ERR_INVALID;
    if (interleave_n < 0 || (interleave_n > 0 && (rows % interleave_n || mul))) return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int cgb, gx, gy, rpb;
    cl_geometry(rows, c, &cgb, &gx, &gy, &rpb);
    hipLaunchKernelGGL(cl_bwd_reduce_kernel, dim3(gx, gy), dim3(BN_THREADS), 0, s, rows, c, cgb, rpb, act, x, dy, mul,
                       stats, scratch, interleave_n);
    float *coef = scratch + (size_t)gy * 2 * c;                // [ca | cb], behind the partials
    hipLaunchKernelGGL(cl_bwd_finalize_kernel, dim3(cdiv(c, FIN_CH)), dim3(FIN_CH * FIN_PL), 0, s, rows, c, gy, training,
                       scratch, stats, bsums, coef);
    hipLaunchKernelGGL(cl_bwd_apply_kernel, dim3(gx, gy), dim3(BN_THREADS), 0, s, rows, c, cgb, rpb, act, x, dy, mul,
                       stats, coef, dx, dmul, interleave_n);
    return pdgn_launch_status();
}

// ---------------------------------------------------------------------------- bilateral weighting, adjoint
// y = act(BN_u(u)) * softmax_slots_permute(act(BN_x(x)))  (softmax_perm.hip: bn_softmax_perm_mul_fwd_kernel).
// Its adjoint chains a BatchNorm backward (u), a slot-softmax backward and a second BatchNorm backward (x); done
// separately that is 8 + 3 + 5 passes over edge-sized tensors with dW and dh materialised in between.  Here a thread
// owns one channel pair of x (= four interleaved channels of u) and walks points: it holds all k slots of the pair,
// so dW, the softmax dot product and dh live in registers.  Two launches (batch sums, then the gradients) read
// x, u, w, dy twice and write dx, du once: 10 passes instead of 16, nothing intermediate in HBM.
#define BW_MAXK 16                      // runtime-k instance: 16 slots in registers (32 needed 62 spilled registers); wider
                                        // neighbourhoods take the unfused route (fused.bilateral_weighting)

template <int KT, bool APPLY>
__global__ __launch_bounds__(BN_THREADS) void bilateral_bwd_kernel(
    long long M, int k_rt, int C, int cpb, int m_per_block, int act, const float *__restrict__ x,
    const float *__restrict__ stats_x, const float *__restrict__ u, const float *__restrict__ stats_u,
    const float *__restrict__ w, const float *__restrict__ dy, float *__restrict__ part_x, float *__restrict__ part_u,
    const float *__restrict__ coef_x, const float *__restrict__ coef_u, float *__restrict__ dx, float *__restrict__ du) {
    constexpr int KM = KT ? KT : BW_MAXK;
    __shared__ float red[12][BN_THREADS];
    const int k = KT ? KT : k_rt, P = k / 2, C2 = C / 2, Cu = 2 * C;
    const int cpl = threadIdx.x % cpb, rlane = threadIdx.x / cpb, rl = BN_THREADS / cpb;
    const int cp = blockIdx.x * cpb + cpl;
    const bool ok = cp < C2;
    const int c = 2 * (ok ? cp : 0);
    const long long m0 = (long long)blockIdx.y * m_per_block, m1 = min(M, m0 + m_per_block);
    const float2 scx = *reinterpret_cast<const float2 *>(stats_x + c), shx = *reinterpret_cast<const float2 *>(stats_x + C + c);
    const float2 mux = *reinterpret_cast<const float2 *>(stats_x + 2 * C + c), isx = *reinterpret_cast<const float2 *>(stats_x + 3 * C + c);
    const float4 scu = *reinterpret_cast<const float4 *>(stats_u + 2 * c), shu = *reinterpret_cast<const float4 *>(stats_u + Cu + 2 * c);
    const float4 muu = *reinterpret_cast<const float4 *>(stats_u + 2 * Cu + 2 * c), isu = *reinterpret_cast<const float4 *>(stats_u + 3 * Cu + 2 * c);
    float2 cax = make_float2(0.f, 0.f), cbx = cax;
    float4 cau = make_float4(0.f, 0.f, 0.f, 0.f), cbu = cau;
    if (APPLY) {
        cax = *reinterpret_cast<const float2 *>(coef_x + c); cbx = *reinterpret_cast<const float2 *>(coef_x + C + c);
        cau = *reinterpret_cast<const float4 *>(coef_u + 2 * c); cbu = *reinterpret_cast<const float4 *>(coef_u + Cu + 2 * c);
    }
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;                  // sx[2] qx[2] su[4] qu[4]
    if (ok)
        for (long long m = m0 + rlane; m < m1; m += rl) {
            const size_t ox = (size_t)m * k * C + c, ou = (size_t)m * k * C + 2 * c;
            float2 xv[KM];
            float4 uv[KM / 2], wv[KM / 2], gv[KM / 2];
#pragma unroll
            for (int s = 0; s < KM; ++s)
                if (KT || s < k) xv[s] = *reinterpret_cast<const float2 *>(x + ox + (size_t)s * C);
#pragma unroll
            for (int p = 0; p < KM / 2; ++p)
                if (KT || p < P) {
                    uv[p] = *reinterpret_cast<const float4 *>(u + ou + (size_t)p * Cu);
                    if (w) wv[p] = *reinterpret_cast<const float4 *>(w + ou + (size_t)p * Cu);
                    gv[p] = *reinterpret_cast<const float4 *>(dy + ou + (size_t)p * Cu);
                }
            if (!w) {
                // w is a function of x alone: recomputed from the slots this thread holds anyway, with the forward's own
                // expression (bn_softmax_perm_mul_fwd_kernel) -- the (M, k/2, 2C) weight tensor is neither written by the
                // forward nor read here: one edge-sized write and two reads per block and iteration less
                float h0[KM], h1[KM];
                float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
                for (int s = 0; s < KM; ++s)
                    if (KT || s < k) {
                        h0[s] = act_fwd(__fmaf_rn(xv[s].x, scx.x, shx.x), act);
                        h1[s] = act_fwd(__fmaf_rn(xv[s].y, scx.y, shx.y), act);
                        mx0 = fmaxf(mx0, h0[s]);
                        mx1 = fmaxf(mx1, h1[s]);
                    }
                float e0 = 0.f, e1 = 0.f;
#pragma unroll
                for (int s = 0; s < KM; ++s)
                    if (KT || s < k) {
                        h0[s] = __expf(h0[s] - mx0);
                        h1[s] = __expf(h1[s] - mx1);
                        e0 += h0[s];
                        e1 += h1[s];
                    }
                const float i0 = 1.0f / e0, i1 = 1.0f / e1;
#pragma unroll
                for (int p = 0; p < KM / 2; ++p)
                    if (KT || p < P) wv[p] = make_float4(h0[p] * i0, h0[P + p] * i0, h1[p] * i1, h1[P + p] * i1);
            }
            float dot0 = 0.f, dot1 = 0.f;
#pragma unroll
            for (int p = 0; p < KM / 2; ++p)
                if (KT || p < P) {
                    const float u4[4] = {uv[p].x, uv[p].y, uv[p].z, uv[p].w}, w4[4] = {wv[p].x, wv[p].y, wv[p].z, wv[p].w};
                    const float g4[4] = {gv[p].x, gv[p].y, gv[p].z, gv[p].w};
                    const float s4[4] = {scu.x, scu.y, scu.z, scu.w}, h4[4] = {shu.x, shu.y, shu.z, shu.w};
                    const float m4[4] = {muu.x, muu.y, muu.z, muu.w}, i4[4] = {isu.x, isu.y, isu.z, isu.w};
                    const float a4[4] = {cau.x, cau.y, cau.z, cau.w}, b4[4] = {cbu.x, cbu.y, cbu.z, cbu.w};
                    float dW[4], o4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float z = __fmaf_rn(u4[j], s4[j], h4[j]);
                        dW[j] = g4[j] * act_fwd(z, act);
                        const float dz = g4[j] * w4[j] * act_grad(z, act);
                        if (APPLY) o4[j] = __fmaf_rn(s4[j], dz, -a4[j]) - b4[j] * u4[j];
                        else { acc[4 + j] += dz; acc[8 + j] = __fmaf_rn(dz, (u4[j] - m4[j]) * i4[j], acc[8 + j]); }
                    }
                    if (APPLY) *reinterpret_cast<float4 *>(du + ou + (size_t)p * Cu) = make_float4(o4[0], o4[1], o4[2], o4[3]);
                    dot0 = __fmaf_rn(w4[0], dW[0], __fmaf_rn(w4[1], dW[1], dot0));
                    dot1 = __fmaf_rn(w4[2], dW[2], __fmaf_rn(w4[3], dW[3], dot1));
                    gv[p] = make_float4(dW[0], dW[1], dW[2], dW[3]);          // keep dW for the softmax adjoint
                }
            // softmax adjoint: slot s = P*j + p of channel c (+1) sits in component j (+2) of row p
#pragma unroll
            for (int s = 0; s < KM; ++s)
                if (KT || s < k) {
                    const int p = s < P ? s : s - P;
                    const bool hi = s >= P;
                    const float w0 = hi ? wv[p].y : wv[p].x, g0 = hi ? gv[p].y : gv[p].x;
                    const float w1 = hi ? wv[p].w : wv[p].z, g1 = hi ? gv[p].w : gv[p].z;
                    const float da0 = w0 * (g0 - dot0), da1 = w1 * (g1 - dot1);
                    const float dz0 = da0 * act_grad(__fmaf_rn(xv[s].x, scx.x, shx.x), act);
                    const float dz1 = da1 * act_grad(__fmaf_rn(xv[s].y, scx.y, shx.y), act);
                    if (APPLY) {
                        float2 o;
                        o.x = __fmaf_rn(scx.x, dz0, -cax.x) - cbx.x * xv[s].x;
                        o.y = __fmaf_rn(scx.y, dz1, -cax.y) - cbx.y * xv[s].y;
                        *reinterpret_cast<float2 *>(dx + ox + (size_t)s * C) = o;
                    } else {
                        acc[0] += dz0; acc[1] += dz1;
                        acc[2] = __fmaf_rn(dz0, (xv[s].x - mux.x) * isx.x, acc[2]);
                        acc[3] = __fmaf_rn(dz1, (xv[s].y - mux.y) * isx.y, acc[3]);
                    }
                }
        }
    if (APPLY) return;
#pragma unroll
    for (int i = 0; i < 12; ++i) red[i][threadIdx.x] = acc[i];
    __syncthreads();
    if (rlane == 0 && ok) {
        for (int j = 1; j < rl; ++j)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] += red[i][j * cpb + cpl];
        float *px = part_x + (size_t)blockIdx.y * 2 * C, *pu = part_u + (size_t)blockIdx.y * 2 * Cu;
        *reinterpret_cast<float2 *>(px + c) = make_float2(acc[0], acc[1]);
        *reinterpret_cast<float2 *>(px + C + c) = make_float2(acc[2], acc[3]);
        *reinterpret_cast<float4 *>(pu + 2 * c) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        *reinterpret_cast<float4 *>(pu + Cu + 2 * c) = make_float4(acc[8], acc[9], acc[10], acc[11]);
    }
}

static void bw_geometry(long long M, int C, int *cpb, int *gx, int *gy, int *mpb) {
    const int c2 = C / 2;
    int p = 1;
    while (p < c2 && p < BN_THREADS) p <<= 1;
    *cpb = p;
    *gx = (c2 + p - 1) / p;
    const int rl = BN_THREADS / p;
    long long want = 1024 / *gx;
    want = want < 1 ? 1 : want;
    long long rows = (M + want - 1) / want;
    rows = rows < (long long)rl * 4 ? (long long)rl * 4 : rows;
    rows = rows > 4096 ? 4096 : rows;                       // bound the fp32 partial sums (x k slots each)
    rows = (rows + rl - 1) / rl * rl;
    *mpb = (int)rows;
    *gy = (int)((M + rows - 1) / rows);
}

// floats of scratch pdgn_bilateral_weighting_backward needs for (m, k, c)
extern "C" long long pdgn_bilateral_scratch_floats(long long m, int k, int c) {
    if (m < 1 || k < 2 || (k & 1) || k > BW_MAXK || c < 2 || (c & 1)) return PDGN_ERR_INVALID;
    int cpb, gx, gy, mpb;
    bw_geometry(m, c, &cpb, &gx, &gy, &mpb);
    return (long long)gy * 6 * c + 6 * c;                   // partials of x (2c) and u (4c) per row block + the two coefficient rows
}

// Adjoint of pdgn_bn_softmax_slots_permute_mul: x (m,k,c), u / w / dy (m,k/2,2c); bsums_x (2c) = [dbeta_x | dgamma_x],
// bsums_u (4c) likewise for BN_u; dx (m,k,c), du (m,k/2,2c).  c even.  w == NULL: the weights are recomputed from x.
extern "C" int pdgn_bilateral_weighting_backward(long long m, int k, int c, int act, int training, const float *x,
                                                 const float *stats_x, const float *u, const float *stats_u,
                                                 const float *w, const float *dy, float *scratch, float *bsums_x,
                                                 float *bsums_u, float *dx, float *du, pdgn_stream_t stream) {
    if (m < 1 || k < 2 || (k & 1) || k > BW_MAXK || c < 4 || (c % 4) || act < 0 || act > 2) return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int cpb, gx, gy, mpb;
    bw_geometry(m, c, &cpb, &gx, &gy, &mpb);
    float *part_x = scratch, *part_u = part_x + (size_t)gy * 2 * c;
    float *coef_x = part_u + (size_t)gy * 4 * c, *coef_u = coef_x + 2 * c;
    const dim3 grid(gx, gy), block(BN_THREADS);
#define BW_LAUNCH(KT, AP)                                                                                              \
    hipLaunchKernelGGL((bilateral_bwd_kernel<KT, AP>), grid, block, 0, s, m, k, c, cpb, mpb, act, x, stats_x, u, stats_u, w, \
                       dy, part_x, part_u, coef_x, coef_u, dx, du)
    if (k == 10) BW_LAUNCH(10, false); else if (k == 4) BW_LAUNCH(4, false); else BW_LAUNCH(0, false);
    hipLaunchKernelGGL(cl_bwd_finalize_kernel, dim3(cdiv(c, FIN_CH)), dim3(FIN_CH * FIN_PL), 0, s, m * k, c, gy, training, part_x,
                       stats_x, bsums_x, coef_x);
    hipLaunchKernelGGL(cl_bwd_finalize_kernel, dim3(cdiv(2 * c, FIN_CH)), dim3(FIN_CH * FIN_PL), 0, s, m * (k / 2), 2 * c, gy,
                       training, part_u, stats_u, bsums_u, coef_u);
    if (k == 10) BW_LAUNCH(10, true); else if (k == 4) BW_LAUNCH(4, true); else BW_LAUNCH(0, true);
#undef BW_LAUNCH
    return pdgn_launch_status();
}

// ---------------------------------------------------------------------------- BN + act + max-pool
// The PointNet-style discriminators end their per-point stack with BatchNorm1d + LeakyReLU +
// MaxPool1d over all points (models/PDGNet_v2.py:886-911 ...).  Only the per-sample maxima leave
// the layer, so the activated (rows x C) tensor is never written: the apply pass reduces to
// (max, argmax) per (sample, channel), and the backward pass is ONE streaming pass
//   dx[r,c] = -ca[c] - cb[c]*x[r,c] + [r == argmax(b,c)] * scale[c]*dz[b,c]
// because dz is non-zero only at the argmax rows (its two batch sums run over B*C values).
#define MP_SPLIT 16

__global__ __launch_bounds__(BN_THREADS) void cl_apply_max_kernel(int N, int C, int cgb, int act,
                                                                  const float *__restrict__ x,
                                                                  const float *__restrict__ stats,
                                                                  float *__restrict__ pmax, int32_t *__restrict__ parg) {
    __shared__ float smax[BN_THREADS][4];
    __shared__ int sarg[BN_THREADS][4];
    const int cgl = threadIdx.x % cgb, rlane = threadIdx.x / cgb, rl = BN_THREADS / cgb;
    const int cgi = blockIdx.x * cgb + cgl;
    const bool cok = cgi * 4 < C;
    const int b = blockIdx.z, sp = blockIdx.y;
    const int per = (N + MP_SPLIT - 1) / MP_SPLIT;
    const int n0 = sp * per, n1 = min(N, n0 + per);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {n0, n0, n0, n0};
    if (cok) {
        float sc[4], sh[4];
        *reinterpret_cast<float4 *>(sc) = *reinterpret_cast<const float4 *>(stats + cgi * 4);
        *reinterpret_cast<float4 *>(sh) = *reinterpret_cast<const float4 *>(stats + C + cgi * 4);
        const float *X = x + ((size_t)b * N) * C + cgi * 4;
        for (int n = n0 + rlane; n < n1; n += rl) {
            float v[4];
            *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(X + (size_t)n * C);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = act_fwd(__fmaf_rn(v[j], sc[j], sh[j]), act);
                if (y > best[j]) { best[j] = y; arg[j] = n; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { smax[threadIdx.x][j] = best[j]; sarg[threadIdx.x][j] = arg[j]; }
    __syncthreads();
    if (rlane == 0 && cok) {
        for (int q = 1; q < rl; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = smax[q * cgb + cgl][j];
                const int a = sarg[q * cgb + cgl][j];
                if (y > best[j] || (y == best[j] && a < arg[j])) { best[j] = y; arg[j] = a; }
            }
        const size_t o = ((size_t)b * MP_SPLIT + sp) * C + cgi * 4;
        *reinterpret_cast<float4 *>(pmax + o) = *reinterpret_cast<float4 *>(best);
        *reinterpret_cast<int4 *>(parg + o) = *reinterpret_cast<int4 *>(arg);
    }
}

__global__ void cl_max_finalize_kernel(int B, int C, const float *__restrict__ pmax, const int32_t *__restrict__ parg,
                                       float *__restrict__ ymax, int32_t *__restrict__ yarg) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * C) return;
    const int b = e / C, c = e % C;
    float best = -INFINITY;
    int arg = 0;
    for (int s = 0; s < MP_SPLIT; ++s) {                       // splits are in row order: first maximum wins
        const float y = pmax[((size_t)b * MP_SPLIT + s) * C + c];
        if (y > best) { best = y; arg = parg[((size_t)b * MP_SPLIT + s) * C + c]; }
    }
    ymax[e] = best;
    yarg[e] = arg;
}

// Statistics AND extremes in one streaming pass: when the BatchNorm's statistics do not arrive from the producer's epilogue, the max-pool
// tail would read x twice (statistics, then apply + max).  act(scale x + shift) is monotone in x with the sign of scale, so the pass
// that sums x (shifted by row 0, as cl_stats_kernel) also keeps, per (sample, split, channel), the largest and the smallest x and their
// rows; the finalisation picks one of them once the scale's sign is known.  (For the discriminators' last layer this is also cheaper
// than statistics in the GEMM's epilogue: 66 us of DPP reductions at 71680 x 1024 x 256 against a pass that is bound by HBM anyway.)
__global__ __launch_bounds__(BN_THREADS) void cl_stats_max_kernel(int N, int C, int cgb, const float *__restrict__ x,
                                                                  float *__restrict__ part, float *__restrict__ pmax,
                                                                  int32_t *__restrict__ pargmax, float *__restrict__ pmin,
                                                                  int32_t *__restrict__ pargmin) {
    __shared__ float4 red[4][BN_THREADS];
    __shared__ int4 redi[2][BN_THREADS];
    const int cgl = threadIdx.x % cgb, rlane = threadIdx.x / cgb, rl = BN_THREADS / cgb;
    const int cgi = blockIdx.x * cgb + cgl;
    const bool cok = cgi * 4 < C;
    const int b = blockIdx.z, sp = blockIdx.y;
    const int per = (N + MP_SPLIT - 1) / MP_SPLIT;
    const int n0 = sp * per, n1 = min(N, n0 + per);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int amx[4] = {n0, n0, n0, n0}, amn[4] = {n0, n0, n0, n0};
    if (cok) {
        float pv[4];
        *reinterpret_cast<float4 *>(pv) = *reinterpret_cast<const float4 *>(x + cgi * 4);            // pivot: row 0 of x
        const float *X = x + ((size_t)b * N) * C + cgi * 4;
        for (int n = n0 + rlane; n < n1; n += rl) {
            float v[4];
            *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(X + (size_t)n * C);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[j] - pv[j];
                s[j] += d;
                q[j] = __fmaf_rn(d, d, q[j]);
                if (v[j] > mx[j]) { mx[j] = v[j]; amx[j] = n; }
                if (v[j] < mn[j]) { mn[j] = v[j]; amn[j] = n; }
            }
        }
    }
    red[0][threadIdx.x] = make_float4(s[0], s[1], s[2], s[3]);
    red[1][threadIdx.x] = make_float4(q[0], q[1], q[2], q[3]);
    red[2][threadIdx.x] = make_float4(mx[0], mx[1], mx[2], mx[3]);
    red[3][threadIdx.x] = make_float4(mn[0], mn[1], mn[2], mn[3]);
    redi[0][threadIdx.x] = make_int4(amx[0], amx[1], amx[2], amx[3]);
    redi[1][threadIdx.x] = make_int4(amn[0], amn[1], amn[2], amn[3]);
    __syncthreads();
    if (rlane || !cok) return;
    for (int k = 1; k < rl; ++k) {
        const int t = k * cgb + cgl;
        const float *a = &red[0][t].x, *bq = &red[1][t].x, *c2 = &red[2][t].x, *d2 = &red[3][t].x;
        const int *i2 = &redi[0][t].x, *j2 = &redi[1][t].x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] += a[j];
            q[j] += bq[j];
            if (c2[j] > mx[j] || (c2[j] == mx[j] && i2[j] < amx[j])) { mx[j] = c2[j]; amx[j] = i2[j]; }
            if (d2[j] < mn[j] || (d2[j] == mn[j] && j2[j] < amn[j])) { mn[j] = d2[j]; amn[j] = j2[j]; }
        }
    }
    const size_t blk = (size_t)b * MP_SPLIT + sp;
    float *P = part + blk * 2 * C;
    *reinterpret_cast<float4 *>(P + cgi * 4) = make_float4(s[0], s[1], s[2], s[3]);
    *reinterpret_cast<float4 *>(P + C + cgi * 4) = make_float4(q[0], q[1], q[2], q[3]);
    const size_t o = blk * C + cgi * 4;
    *reinterpret_cast<float4 *>(pmax + o) = make_float4(mx[0], mx[1], mx[2], mx[3]);
    *reinterpret_cast<int4 *>(pargmax + o) = make_int4(amx[0], amx[1], amx[2], amx[3]);
    *reinterpret_cast<float4 *>(pmin + o) = make_float4(mn[0], mn[1], mn[2], mn[3]);
    *reinterpret_cast<int4 *>(pargmin + o) = make_int4(amn[0], amn[1], amn[2], amn[3]);
}

// per (sample, channel): the extreme the scale's sign asks for over the MP_SPLIT splits (in row order: the first one wins)
__global__ void cl_max_pick_kernel(int B, int C, int act, const float *__restrict__ stats, const float *__restrict__ pmax,
                                   const int32_t *__restrict__ pargmax, const float *__restrict__ pmin,
                                   const int32_t *__restrict__ pargmin, float *__restrict__ ymax, int32_t *__restrict__ yarg) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * C) return;
    const int b = e / C, c = e - b * C;
    const float sc = stats[c], sh = stats[C + c];
    const bool up = sc >= 0.f;
    const float *pv = up ? pmax : pmin;
    const int32_t *pa = up ? pargmax : pargmin;
    float best = pv[((size_t)b * MP_SPLIT) * C + c];
    int arg = pa[((size_t)b * MP_SPLIT) * C + c];
    for (int s = 1; s < MP_SPLIT; ++s) {
        const float v = pv[((size_t)b * MP_SPLIT + s) * C + c];
        if (up ? v > best : v < best) { best = v; arg = pa[((size_t)b * MP_SPLIT + s) * C + c]; }
    }
    ymax[e] = act_fwd(__fmaf_rn(best, sc, sh), act);
    yarg[e] = arg;
}

// per channel: dz[b] = dout[b]*act'(z at the argmax row); bsums = [sum_b dz | sum_b dz*xhat]; coef = [ca | cb].
// A block is 16 channels x 16 sample lanes (the B gathers of a channel are dependent-latency loads: one thread per channel
// walking all B of them took 105 us at B = 35, C = 1024).
__global__ __launch_bounds__(256) void cl_max_bwd_sums_kernel(int B, int N, int C, int act, int training,
                                                              const float *__restrict__ x, const float *__restrict__ dout,
                                                              const int32_t *__restrict__ yarg, const float *__restrict__ stats,
                                                              float *__restrict__ dz, float *__restrict__ bsums,
                                                              float *__restrict__ coef, float *__restrict__ zero, int nzero) {
    __shared__ double p1[16][17], p2[16][17];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nzero; e += gridDim.x * 256) zero[e] = 0.f;   // side job (cf_gram's output)
    const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0, s2 = 0;
    float sc = 0.f, mu = 0.f, is = 0.f;
    if (c < C) {
        sc = stats[c];
        mu = stats[2 * C + c];
        is = stats[3 * C + c];
        const float sh = stats[C + c];
        for (int b = bl; b < B; b += 16) {
            const int n = yarg[b * C + c];
            const float v = x[((size_t)b * N + n) * C + c];
            const float g = dout[b * C + c] * act_grad(__fmaf_rn(v, sc, sh), act);
            dz[b * C + c] = g;
            s1 += g;
            s2 += (double)g * (double)((v - mu) * is);
        }
    }
    p1[bl][cl] = s1;
    p2[bl][cl] = s2;
    __syncthreads();
    if (bl || c >= C) return;
    for (int q = 1; q < 16; ++q) { s1 += p1[q][cl]; s2 += p2[q][cl]; }
    bsums[c] = (float)s1;
    bsums[C + c] = (float)s2;
    float ca = 0.f, cb = 0.f;
    if (training) {
        const double R = (double)B * (double)N;
        cb = sc * is * (float)(s2 / R);
        ca = sc * (float)(s1 / R) - cb * mu;
    }
    coef[c] = ca;
    coef[C + c] = cb;
}

__global__ __launch_bounds__(BN_THREADS) void cl_max_bwd_apply_kernel(int N, int C, int cgb, const float *__restrict__ x,
                                                                      const float *__restrict__ stats,
                                                                      const float *__restrict__ coef,
                                                                      const float *__restrict__ dz,
                                                                      const int32_t *__restrict__ yarg,
                                                                      float *__restrict__ dx) {
    const int cgl = threadIdx.x % cgb, rlane = threadIdx.x / cgb, rl = BN_THREADS / cgb;
    const int cgi = blockIdx.x * cgb + cgl;
    if (cgi * 4 >= C) return;
    const int b = blockIdx.z, sp = blockIdx.y;
    const int per = (N + MP_SPLIT - 1) / MP_SPLIT;
    const int n0 = sp * per, n1 = min(N, n0 + per);
    float sc[4], ca[4], cb[4], g[4];
    int arg[4];
    *reinterpret_cast<float4 *>(sc) = *reinterpret_cast<const float4 *>(stats + cgi * 4);
    *reinterpret_cast<float4 *>(ca) = *reinterpret_cast<const float4 *>(coef + cgi * 4);
    *reinterpret_cast<float4 *>(cb) = *reinterpret_cast<const float4 *>(coef + C + cgi * 4);
    *reinterpret_cast<float4 *>(g) = *reinterpret_cast<const float4 *>(dz + (size_t)b * C + cgi * 4);
    *reinterpret_cast<int4 *>(arg) = *reinterpret_cast<const int4 *>(yarg + (size_t)b * C + cgi * 4);
    const size_t base = ((size_t)b * N) * C + cgi * 4;
    for (int n = n0 + rlane; n < n1; n += rl) {
        float v[4], o[4];
        *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(x + base + (size_t)n * C);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (n == arg[j] ? sc[j] * g[j] : 0.f) - ca[j] - cb[j] * v[j];
        *reinterpret_cast<float4 *>(dx + base + (size_t)n * C) = *reinterpret_cast<float4 *>(o);
    }
}

static void mp_geometry(int C, int *cgb, int *gx) {
    const int cg = C / 4;
    int p = 1;
    while (p < cg && p < 64) p <<= 1;                          // <= 64 column groups: >= 4 row lanes per block
    *cgb = p;
    *gx = (cg + p - 1) / p;
}

// ymax/yarg (b,c) = max / argmax over the n rows of each sample of act(x*scale + shift); x (b*n, c).
// scratch: b*MP_SPLIT*c floats + as many int32 (use pdgn_bn_maxpool_scratch_floats).
extern "C" long long pdgn_bn_maxpool_scratch_floats(int b, int c) { return 2LL * b * MP_SPLIT * c; }

extern "C" int pdgn_bn_act_maxpool(int b, int n, int c, int act, const float *x, const float *stats, float *scratch,
                                   float *ymax, int32_t *yarg, pdgn_stream_t stream) {
    if (b < 1 || n < 1 || c < 4 || c % 4 || act < 0 || act > 2 || b > 65535) return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int cgb, gx;
    mp_geometry(c, &cgb, &gx);
    float *pmax = scratch;
    int32_t *parg = reinterpret_cast<int32_t *>(scratch + (size_t)b * MP_SPLIT * c);
    hipLaunchKernelGGL(cl_apply_max_kernel, dim3(gx, MP_SPLIT, b), dim3(BN_THREADS), 0, s, n, c, cgb, act, x, stats, pmax, parg);
    hipLaunchKernelGGL(cl_max_finalize_kernel, dim3(cdiv((long long)b * c, 256)), dim3(256), 0, s, b, c, pmax, parg, ymax, yarg);
    return pdgn_launch_status();
}

// The same from x alone, in training mode: batch statistics (+ running-statistics update, as pdgn_bn_stats) and the pooled output
// in ONE pass over x.  stats (4c) out.  scratch: pdgn_bn_stats_maxpool_scratch_floats(b, c) floats.
extern "C" long long pdgn_bn_stats_maxpool_scratch_floats(int b, int c) { return 6LL * b * MP_SPLIT * c; }

extern "C" int pdgn_bn_stats_act_maxpool(int b, int n, int c, int act, float eps, float momentum, const float *x, const float *gamma,
                                         const float *beta, const float *pre_bias, float *running_mean, float *running_var,
                                         float *scratch, float *stats, float *ymax, int32_t *yarg, pdgn_stream_t stream) {
    if (b < 1 || n < 1 || c < 4 || c % 4 || act < 0 || act > 2 || b > 65535) return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int cgb, gx;
    mp_geometry(c, &cgb, &gx);
    const size_t bc = (size_t)b * MP_SPLIT * c;
    float *part = scratch, *pmax = part + 2 * bc, *pmin = pmax + 2 * bc;
    int32_t *pargmax = reinterpret_cast<int32_t *>(pmax + bc), *pargmin = reinterpret_cast<int32_t *>(pmin + bc);
    hipLaunchKernelGGL(cl_stats_max_kernel, dim3(gx, MP_SPLIT, b), dim3(BN_THREADS), 0, s, n, c, cgb, x, part, pmax, pargmax, pmin,
                       pargmin);
    hipLaunchKernelGGL(cl_finalize_kernel, dim3(cdiv(c, FIN_CH)), dim3(FIN_CH * FIN_PL), 0, s, (long long)b * n, c, b * MP_SPLIT, eps,
                       momentum, part, gamma, beta, pre_bias, running_mean, running_var, stats, x);
    hipLaunchKernelGGL(cl_max_pick_kernel, dim3(cdiv((long long)b * c, 256)), dim3(256), 0, s, b, c, act, stats, pmax, pargmax, pmin,
                       pargmin, ymax, yarg);
    return pdgn_launch_status();
}

// dx (b*n, c) from dout (b,c); bsums (2c) = [dbeta | dgamma]; scratch: b*c + 2c floats.
extern "C" int pdgn_bn_act_maxpool_backward(int b, int n, int c, int act, int training, const float *x,
                                            const float *dout, const int32_t *yarg, const float *stats, float *scratch,
                                            float *bsums, float *dx, pdgn_stream_t stream) {
    if (b < 1 || n < 1 || c < 4 || c % 4 || act < 0 || act > 2 || b > 65535) return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    float *dz = scratch, *coef = scratch + (size_t)b * c;
    hipLaunchKernelGGL(cl_max_bwd_sums_kernel, dim3(cdiv(c, 16)), dim3(256), 0, s, b, n, c, act, training, x, dout, yarg,
                       stats, dz, bsums, coef, nullptr, 0);
    int cgb, gx;
    mp_geometry(c, &cgb, &gx);
    hipLaunchKernelGGL(cl_max_bwd_apply_kernel, dim3(gx, MP_SPLIT, b), dim3(BN_THREADS), 0, s, n, c, cgb, x, stats, coef, dz,
                       yarg, dx);
    return pdgn_launch_status();
}

// ---------------------------------------------------------------------------- dense -> BN -> act -> max-pool: input gradient
// The last per-point layer of a discriminator with FROZEN parameters (the generator's update, models/PDGNet_v2.py:330-352:
// only the gradient with respect to the points is wanted).  With x = h W^T (rows x C; the Conv1d bias cancels in the batch
// statistics) and the adjoint of the pooled BatchNorm above,  dx = S - 1 ca^T - x diag(cb)  (S non-zero at the B*C arg-max
// entries only), the input gradient needs neither the dense (rows x C) dx nor a (rows x C x K) product:
//     dh = dx W = S W  -  1 (ca^T W)  -  h (W^T diag(cb) W)
// = one (rows x K x K) product with a bias epilogue (G = -W^T diag(cb) W and v = -ca^T W are K x K and K) plus a scatter of
// B*C scaled rows of W.  At D4 (rows = 71680, C = 1024, K = 256): 9.4 GFLOP and 73 MB written instead of 37.6 GFLOP behind a
// 293 MB dx that is written and read back.
#define CF_TILE 32         // G tile edge of a block (each thread 4 x 4 outputs)
#define CF_CSPLIT 8        // channel groups across blocks (x 4 wavefronts a block = 32 channel slices)
#define CF_RSPLIT 32       // blocks per sample in the scatter

// Gneg (K x K) -= sum_c cb[c] W[c,i] W[c,j]; vneg (K) -= sum_c ca[c] W[c,j]; both ZERO on entry (cl_max_bwd_sums_kernel's side
// job).  grid (ceil(K/32), ceil(K/32), CF_CSPLIT) x 4 wavefronts, each a 4 x 4 block of G per thread over its slice of the
// channels: float4 loads of W's rows, sixteen rows in flight (branch-free: a load behind a branch is a round trip of its own),
// joined in LDS and added to G with CF_CSPLIT-way atomics.
__global__ __launch_bounds__(256) void cf_gram_kernel(int C, int K, const float *__restrict__ W, int ldw,
                                                      const float *__restrict__ coef, float *__restrict__ Gneg,
                                                      float *__restrict__ vneg) {
    __shared__ float part[3][64][20];
    const int t = threadIdx.x, o = t & 63, sl = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i = blockIdx.y * CF_TILE + (o >> 3) * 4, j = blockIdx.x * CF_TILE + (o & 7) * 4;
    const bool ok = i < K && j < K, do_v = blockIdx.y == 0 && (o >> 3) == 0 && j < K;
    const int ii = min(i, K - 4), jj = min(j, K - 4);
    const int nsl = CF_CSPLIT * 4, per = (C + nsl - 1) / nsl;
    const int c0 = (blockIdx.z * 4 + sl) * per, c1 = min(C, c0 + per);
    const float *cb = coef + C;
    float acc[4][4] = {}, av[4] = {};
    for (int c = c0; c < c1; c += 8) {
        float4 wi[8], wj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float *w = W + (size_t)min(c + u, c1 - 1) * ldw;
            wi[u] = *reinterpret_cast<const float4 *>(w + ii);
            wj[u] = *reinterpret_cast<const float4 *>(w + jj);
        }
        __builtin_amdgcn_sched_barrier(0);                           // all sixteen loads issued before the first use
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cc = min(c + u, c1 - 1);
            const float live = c + u < c1 ? 1.f : 0.f;
            const float s = cb[cc] * live, a = coef[cc] * live;
            const float x[4] = {wi[u].x * s, wi[u].y * s, wi[u].z * s, wi[u].w * s};
            const float y[4] = {wj[u].x, wj[u].y, wj[u].z, wj[u].w};
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[p][q] = __fmaf_rn(x[p], y[q], acc[p][q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = __fmaf_rn(a, y[q], av[q]);
        }
    }
    if (sl) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) part[sl - 1][o][p * 4 + q] = acc[p][q];
#pragma unroll
        for (int q = 0; q < 4; ++q) part[sl - 1][o][16 + q] = av[q];
    }
    __syncthreads();
    if (sl) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[p][q] += part[w][o][p * 4 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] += part[w][o][16 + q];
    }
    if (ok)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(&Gneg[(size_t)(i + p) * K + j + q], -acc[p][q]);
    if (do_v)
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(&vneg[j + q], -av[q]);
}

// dh[b*N + n, :] += sum over the channels c whose arg-max row in sample b is n of scale[c] dz[b,c] W[c, :]   (K <= 256).
// The arg-max rows of a sample are few and repeat (the critical points of a PointNet): each block compacts them in LDS
// (mark, scan), and wavefront w of block s takes every (CF_RSPLIT*4)-th distinct row: it finds the row's channels by ballot
// over the LDS copy of the sample's arg-max list, sums their W rows in registers (one float4 per lane = a whole row) and adds
// the result to dh with one read-modify-write -- no atomics, every access a full row.
// LDS: uint16 table[N] | uint16 rows[min(N,C)] | uint16 args[C] | (pad) | uint32 cnt[256] | float cf[C]
__global__ __launch_bounds__(256) void cf_scatter_kernel(int N, int C, int K, const float *__restrict__ W, int ldw,
                                                         const float *__restrict__ stats, const float *__restrict__ dz,
                                                         const int32_t *__restrict__ yarg, float *__restrict__ dh) {
    extern __shared__ unsigned char cf_lds[];
    __shared__ unsigned int ndist;
    const int maxd = min(N, C);
    unsigned short *table = reinterpret_cast<unsigned short *>(cf_lds);
    unsigned short *rows = table + N;
    unsigned short *args = rows + maxd;
    unsigned int *cnt = reinterpret_cast<unsigned int *>(cf_lds + (((size_t)(N + maxd + C) * 2 + 15) & ~(size_t)15));
    float *cf = reinterpret_cast<float *>(cnt + 256);
    const int t = threadIdx.x, b = blockIdx.y;
    const int32_t *arg = yarg + (size_t)b * C;
    for (int n = t; n < N; n += 256) table[n] = 0;
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        const int a = arg[c];
        table[a] = 1;
        args[c] = (unsigned short)a;
        cf[c] = stats[c] * dz[(size_t)b * C + c];
    }
    __syncthreads();
    const int per = (N + 255) / 256, lo = t * per, hi = min(N, lo + per);   // thread t owns the entries [lo, hi)
    unsigned int mine = 0;
    for (int n = lo; n < hi; ++n) mine += table[n];
    cnt[t] = mine;
    __syncthreads();
    if (t < 64) {                                                    // exclusive scan of 256 counts by one wavefront
        const unsigned int v0 = cnt[4 * t], v1 = cnt[4 * t + 1], v2 = cnt[4 * t + 2], v3 = cnt[4 * t + 3];
        const unsigned int s = v0 + v1 + v2 + v3;
        unsigned int incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int up = __shfl_up(incl, d, 64);
            if (t >= d) incl += up;
        }
        const unsigned int ex = incl - s;
        cnt[4 * t] = ex;
        cnt[4 * t + 1] = ex + v0;
        cnt[4 * t + 2] = ex + v0 + v1;
        cnt[4 * t + 3] = ex + v0 + v1 + v2;
    }
    __syncthreads();
    unsigned int at = cnt[t];
    for (int n = lo; n < hi; ++n)
        if (table[n]) rows[at++] = (unsigned short)n;
    if (t == 255) ndist = at;                                       // thread 255's running index ends at the total
    __syncthreads();
    const int nd = (int)ndist, lane = t & 63, wv = t >> 6;
    const bool lok = lane * 4 < K;
    const int col = min(lane * 4, K - 4);
    for (int r = blockIdx.x * 4 + wv; r < nd; r += CF_RSPLIT * 4) {
        const int n = rows[r];
        float *dst = dh + ((size_t)b * N + n) * K + col;
        const float4 cur = *reinterpret_cast<const float4 *>(dst);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c0 = 0; c0 < C; c0 += 64) {
            unsigned long long m = __ballot(c0 + lane < C && args[min(c0 + lane, C - 1)] == n);
            while (m) {
                const int c = c0 + __builtin_ctzll(m);
                m &= m - 1;
                const float g = cf[c];
                const float4 w = *reinterpret_cast<const float4 *>(W + (size_t)c * ldw + col);
                acc.x = __fmaf_rn(g, w.x, acc.x);
                acc.y = __fmaf_rn(g, w.y, acc.y);
                acc.z = __fmaf_rn(g, w.z, acc.z);
                acc.w = __fmaf_rn(g, w.w, acc.w);
            }
        }
        if (lok) *reinterpret_cast<float4 *>(dst) = make_float4(cur.x + acc.x, cur.y + acc.y, cur.z + acc.z, cur.w + acc.w);
    }
}

static size_t cf_scatter_lds(int n, int c) {
    const int maxd = n < c ? n : c;
    return (((size_t)(n + maxd + c) * 2 + 15) & ~(size_t)15) + 256 * 4 + (size_t)c * 4;
}

// column sums of h (rows x K, K % 4 == 0, pitch ldh) added into hs (zero on entry).  1024 threads a workgroup (K / 4 column groups x
// row lanes, eight loads in flight each) and at most 128 workgroups: every workgroup ends with K atomic adds onto the SAME K floats --
// with 1024 workgroups those were 1024-deep chains per address and the kernel took 110 us inside the iteration for 73 MB.
#define CFS_THREADS 1024
__global__ __launch_bounds__(CFS_THREADS) void cf_colsum_kernel(long long rows, int K, const float *__restrict__ h, int ldh,
                                                                float *__restrict__ hs) {
    __shared__ float4 red[CFS_THREADS];
    const int kg = K / 4, cg = threadIdx.x % kg, rl = threadIdx.x / kg, nrl = CFS_THREADS / kg;   // kg divides 1024 (K a power of two <= 256)
    const long long per = (rows + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long r = r0 + rl; r < r1; r += 8 * nrl) {            // eight independent loads a round
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long rr = r + (long long)u * nrl;
            v[u] = rr < r1 ? *reinterpret_cast<const float4 *>(h + rr * ldh + cg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl) return;
    for (int q = 1; q < nrl; ++q) {
        const float4 v = red[q * kg + cg];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    atomicAdd(&hs[cg * 4], acc.x); atomicAdd(&hs[cg * 4 + 1], acc.y); atomicAdd(&hs[cg * 4 + 2], acc.z); atomicAdd(&hs[cg * 4 + 3], acc.w);
}

// dW[c, :] = sum_b scale[c] dz[b,c] h[b*N + arg[b,c], :]  -  ca[c] hs  -  cb[c] T[c, :]      (T = W h^T h; K <= 256)
// one wavefront per channel, a float4 of the row per lane; the B row gathers of a channel are independent loads.
__global__ __launch_bounds__(256) void cf_dw_kernel(int B, int N, int C, int K, const float *__restrict__ h, int ldh,
                                                    const int32_t *__restrict__ yarg, const float *__restrict__ stats,
                                                    const float *__restrict__ dz, const float *__restrict__ coef,
                                                    const float *__restrict__ hs, const float *__restrict__ T,
                                                    float *__restrict__ dW, int lddw) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C || lane * 4 >= K) return;
    const int col = lane * 4;
    const float sc = stats[c];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b0 = 0; b0 < B; b0 += 8) {
        float4 v[8];
        float g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = min(b0 + u, B - 1);
            g[u] = b0 + u < B ? sc * dz[(size_t)b * C + c] : 0.f;
            v[u] = *reinterpret_cast<const float4 *>(h + ((size_t)b * N + yarg[(size_t)b * C + c]) * ldh + col);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc.x = __fmaf_rn(g[u], v[u].x, acc.x);
            acc.y = __fmaf_rn(g[u], v[u].y, acc.y);
            acc.z = __fmaf_rn(g[u], v[u].z, acc.z);
            acc.w = __fmaf_rn(g[u], v[u].w, acc.w);
        }
    }
    const float ca = coef[c], cb = coef[C + c];
    const float4 s = *reinterpret_cast<const float4 *>(hs + col), t = *reinterpret_cast<const float4 *>(T + (size_t)c * K + col);
    *reinterpret_cast<float4 *>(dW + (size_t)c * lddw + col) =
        make_float4(acc.x - ca * s.x - cb * t.x, acc.y - ca * s.y - cb * t.y, acc.z - ca * s.z - cb * t.z, acc.w - ca * s.w - cb * t.w);
}

extern "C" long long pdgn_dense_bn_maxpool_backward_scratch(int b, int c, int k) {
    return (long long)b * c + 2LL * c + 2LL * ((long long)k * k + k) + (long long)c * k;
}

// The whole adjoint of [dense layer -> BatchNorm (training statistics) -> act -> max-pool over the n points] without the dense
// (b*n, c) gradient.  x (b*n, c) = the layer's raw output h W^T, yarg / stats as saved by pdgn_bn_act_maxpool, dout (b, c),
// h (b*n, k) pitch ldh, W (c, k) pitch ldw.  Outputs (each may be NULL): dh (b*n, k) contiguous, dW (c, k) pitch lddw,
// bsums (2c) = [dbeta | dgamma].  scratch: pdgn_dense_bn_maxpool_backward_scratch floats.
//   dx = S - 1 ca^T - x diag(cb)   =>   dh = S W - 1 (ca^T W) - h (W^T diag(cb) W),   dW = S^T h - ca (1^T h) - diag(cb) W (h^T h)
extern "C" int pdgn_dense_bn_maxpool_backward(int b, int n, int c, int k, int act, const float *x, const float *dout,
                                              const int32_t *yarg, const float *stats, const float *h, int ldh, const float *W,
                                              int ldw, float *scratch, float *dh, float *dW, int lddw, float *bsums,
                                              pdgn_stream_t stream) {
    if (b < 1 || n < 1 || n > 65535 || c < 4 || c % 4 || k < 4 || k % 4 || k > 256 || ldw % 4 || ((uintptr_t)W & 15) ||
        ((uintptr_t)dh & 15) || act < 0 || act > 2 || b > 65535 || cf_scatter_lds(n, c) > 64 * 1024)
        return PDGN_ERR_INVALID;
    if (dW && (k < 16 || (k & (k - 1)) || ldh % 4 || lddw % 4 || ((uintptr_t)h & 15) || ((uintptr_t)dW & 15)))
        return PDGN_ERR_INVALID;                                    // (the column-sum kernel's geometry: k a power of two)
    hipStream_t s = (hipStream_t)stream;
    const size_t kk = (size_t)k * k;
    float *dz = scratch, *coef = dz + (size_t)b * c, *G = coef + 2 * (size_t)c, *v = G + kk, *H = v + k, *hs = H + kk, *T = hs + k;
    float *bs = bsums ? bsums : T;                                  // (T is written later: a scratch home for unwanted sums)
    hipLaunchKernelGGL(cl_max_bwd_sums_kernel, dim3(cdiv(c, 16)), dim3(256), 0, s, b, n, c, act, 1, x, dout, yarg, stats, dz, bs,
                       coef, G, (int)(2 * (kk + k)));
    int rc = pdgn_launch_status();
    if (rc) return rc;
    if (dh) {
        const int gt = cdiv(k, CF_TILE);
        hipLaunchKernelGGL(cf_gram_kernel, dim3(gt, gt, CF_CSPLIT), dim3(256), 0, s, c, k, W, ldw, coef, G, v);
        if ((rc = pdgn_launch_status())) return rc;
        if ((rc = pdgn_gemm_nt((long long)b * n, k, k, h, ldh, G, k, v, nullptr, 0, dh, k, nullptr, stream))) return rc;
        hipLaunchKernelGGL(cf_scatter_kernel, dim3(CF_RSPLIT, b), dim3(256), cf_scatter_lds(n, c), s, n, c, k, W, ldw, stats, dz,
                           yarg, dh);
        if ((rc = pdgn_launch_status())) return rc;
    }
    if (dW) {
        const long long rows = (long long)b * n;
        hipLaunchKernelGGL(cf_colsum_kernel, dim3(rows >= 32768 ? 128 : (rows >= 8192 ? 64 : 16)), dim3(CFS_THREADS), 0, s, rows, k, h, ldh, hs);
        if ((rc = pdgn_launch_status())) return rc;
        if ((rc = pdgn_gemm_tn_big(rows, k, k, h, ldh, h, ldh, H, 1, stream))) return rc;          // H = h^T h (zeroed above)
        if ((rc = pdgn_gemm_nt(c, k, k, W, ldw, H, k, nullptr, nullptr, 0, T, k, nullptr, stream))) return rc;   // T = W H
        hipLaunchKernelGGL(cf_dw_kernel, dim3(cdiv(c, 4)), dim3(256), 0, s, b, n, c, k, h, ldh, yarg, stats, dz, coef, hs, T, dW, lddw);
        rc = pdgn_launch_status();
    }
    return rc;
}

extern "C" long long pdgn_dense_bn_maxpool_input_grad_scratch(int b, int c, int k) {
    return pdgn_dense_bn_maxpool_backward_scratch(b, c, k);
}

// The input gradient alone (frozen parameters): pdgn_dense_bn_maxpool_backward with dW = bsums = NULL.
extern "C" int pdgn_dense_bn_maxpool_input_grad(int b, int n, int c, int k, int act, const float *x, const float *dout,
                                                const int32_t *yarg, const float *stats, const float *h, int ldh,
                                                const float *W, int ldw, float *scratch, float *dh, pdgn_stream_t stream) {
    return pdgn_dense_bn_maxpool_backward(b, n, c, k, act, x, dout, yarg, stats, h, ldh, W, ldw, scratch, dh, nullptr, 4, nullptr,
                                          stream);
}

// ---------------------------------------------------------------------------- column sums per group of rows
// out[g, c] += sum over the rows r of group g (rows g*group_rows .. ) of x[r, c]  (out zero on entry; c a power of two, 16 .. 1024;
// x pitch ldx, 16-byte aligned rows).  The bias gradients that are not analytically zero (layers without a BatchNorm behind them:
// the heads' convolutions, the per-sample biases of the re-associated edge convolutions, models/PDGNet_v2.py:835-862, :604-618)
// and the per-sample sums of the heads' adjoint -- 21 at::native reduce kernels per iteration on the issuing stream before.
__global__ __launch_bounds__(256) void group_colsum_kernel(long long group_rows, int C, const float *__restrict__ x, long long ldx,
                                                           float *__restrict__ out) {
    __shared__ float4 red[256];
    const int kg = C / 4, per_pass = kg < 256 ? kg : 256;            // column groups handled per pass of the block
    const int cgl = threadIdx.x % per_pass, rl = threadIdx.x / per_pass, nrl = 256 / per_pass;
    const long long g = blockIdx.y;
    const long long per = (group_rows + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(group_rows, r0 + per);
    const float *X = x + g * group_rows * ldx;
    for (int c0 = 0; c0 < kg; c0 += per_pass) {
        const int cg = c0 + cgl;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long r = r0 + rl; r < r1; r += 8 * nrl) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long rr = r + (long long)u * nrl;
                v[u] = rr < r1 ? *reinterpret_cast<const float4 *>(X + rr * ldx + cg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        __syncthreads();
        red[threadIdx.x] = acc;
        __syncthreads();
        if (rl == 0) {
            for (int q = 1; q < nrl; ++q) {
                const float4 v = red[q * per_pass + cgl];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            float *o = out + g * C + cg * 4;
            if (gridDim.x == 1) *reinterpret_cast<float4 *>(o) = acc;
            else { atomicAdd(o, acc.x); atomicAdd(o + 1, acc.y); atomicAdd(o + 2, acc.z); atomicAdd(o + 3, acc.w); }
        }
    }
}

// out (groups, c) contiguous, ZERO on entry when a group is split over several workgroups (always pass it zeroed).
extern "C" int pdgn_group_colsum(long long groups, long long group_rows, int c, const float *x, long long ldx, float *out,
                                 pdgn_stream_t stream) {
    if (groups < 1 || group_rows < 1 || c < 16 || c > 1024 || (c & (c - 1)) || ldx < c || ldx % 4 || ((uintptr_t)x & 15) ||
        ((uintptr_t)out & 15) || groups > 65535)
        return PDGN_ERR_INVALID;
    const int kg = c / 4, nrl = kg < 256 ? 256 / kg : 1;
    long long splits = 1024 / groups;                               // ~1024 workgroups, but <= 128 a group: their sums meet in atomics on
    splits = splits > 128 ? 128 : splits;                           // the same C floats (1024-deep chains per address were the kernel's time)
    const long long max_splits = group_rows / (8LL * nrl * 8) > 0 ? group_rows / (8LL * nrl * 8) : 1;
    splits = splits < 1 ? 1 : (splits > max_splits ? max_splits : splits);
    hipLaunchKernelGGL(group_colsum_kernel, dim3((unsigned)splits, (unsigned)groups), dim3(256), 0, (hipStream_t)stream, group_rows, c, x,
                       ldx, out);
    return pdgn_launch_status();
}
