// wgs.hip -- "window gather-sum": the gather half of the re-associated edge convolutions.
//
// The reference convolves materialised edge tensors e[b,c,n,s] = [x_n, x_idx(n,s) - x_n]
// (models/PDGNet_v2.py:462-477) with 1xT kernels sliding over the k neighbour slots
// (inte_conv_hk :562/:622, conv2 :559/:602, conv_fea :609).  Convolution is linear, so
//     sum_c sum_t W[o,c,t] e[c,n,p+t]  =  sum_t (W2_t X)[o, idx(n,p+t)]  +  ((sum_t W1_t - W2_t) X)[o,n]
// where Y = [W2_0 X | .. | W2_{T-1} X | Wc X] is ONE dense per-point GEMM (MFMA) and what remains is
//     out[b,n,p,c] = bias[b*bias_bstride + c] + Y[b,n,offc+c] + sum_{t<T} Y[b, idx[b,n,p+t], off + t*C + c]
// -- this kernel.  It is a pure row gather (HBM/L2-bound): point-major rows, channels contiguous,
// float4 per lane, the neighbour index wave-uniform.
// Backward scatters dout rows back onto dY with contiguous 256-B wave atomics.
#include <stdlib.h>
#include "common.h"
#include "bn_geom.h"

#define WGS_THREADS 256

template <int VEC>
__global__ __launch_bounds__(WGS_THREADS) void wgs_fwd_kernel(
    long long total, int n, int k, int ldy, int T, int P, int CV, int off, int offc,
    const float *__restrict__ Y, const int32_t *__restrict__ idx, const float *__restrict__ bias,
    int bias_bstride, float *__restrict__ out) {
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
    if (bias) acc = *reinterpret_cast<const vec_t *>(bias + (bn / n) * bias_bstride + c);
    else for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (offc >= 0) acc += *reinterpret_cast<const vec_t *>(Y + bn * ldy + offc + c);
    for (int t = 0; t < T; ++t) {
        const long long row = b0 + I[t];
        acc += *reinterpret_cast<const vec_t *>(Y + row * ldy + off + t * CV * VEC + c);
    }
    *reinterpret_cast<vec_t *>(out + e * VEC) = acc;
}

// Cache-aware variant of the float4 path.  Every Y row segment (j, t) is read by each of the ~P windows that hold j
// in slot p+t, from query points scattered over the sample: 5.3 GB of loads for 0.9 GB of Y at stage 4, and with
// workgroups walking whole rows (C = 1024 channels = 4 KB) a sample's working set (25 MB) never fits an XCD's 4 MB L2.
// Here a TASK is (sample, chunk of WGS_CW float4 = 64 channels): 1.5 MB of Y segments + 0.3 MB of centre columns, read
// by n*P output rows.  Workgroups are dealt round-robin to the 8 XCDs, so block i runs on XCD i%8: all blocks of a task
// take ids of one residue class, consecutive in launch order, and the re-reads hit that XCD's L2.
#define WGS_CW 16          // default chunk width in float4 (64 channels); kernels are templated on it
#define WGS_XU 4          // rows in flight per thread: all their index loads, then all their row loads
// Addressing: everything a task touches lies in ONE sample's slabs of Y / idx / out (tens of MB), so the kernels of this
// mapping use raw buffer descriptors (scalar base per sample) with 32-bit byte offsets built from 24-bit multiplies, and
// the row -> (point, position) split is one multiply-high by a host-computed reciprocal.  (The first version computed a
// 64-bit address per load: 430 vector instructions per thread for 33 loads, 75 of them quarter-rate v_mul_lo_u32 --
// the address arithmetic, not the memory system, set the kernel's time.)
typedef float wgs_vec_t __attribute__((ext_vector_type(4)));
typedef unsigned wgs_u4_t __attribute__((ext_vector_type(4)));
#define WGS_OOB 0x80000000u                              // past every slab (< 2^31 bytes, checked by the launchers)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wgs_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ wgs_vec_t wgs_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(wgs_vec_t, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ unsigned wgs_mul24(unsigned a, unsigned b) { return __umul24(a, b); }
__device__ __forceinline__ unsigned wgs_div(unsigned x, int d, unsigned magic) { return d == 1 ? x : __umulhi(x, magic); }
static unsigned wgs_magic(int d) { return d == 1 ? 0u : (unsigned)(0x100000000ULL / (unsigned)d) + 1u; }   // x / d == umulhi(x, magic) for x * d < 2^32 (d > 1)
static bool wgs_slabs_ok(int n, int k, int ldy, int P, int C) {
    return (long long)n * ldy * 4 < 0x7fffffffLL && (long long)n * P * C * 4 < 0x7fffffffLL && (long long)n * k * 4 < 0x7fffffffLL &&
           (long long)n * P * P < 0x100000000LL && ldy * 4LL < (1 << 24) && (long long)P * C * 4 < (1 << 24) && n < (1 << 24) &&
           (long long)n * P < (1 << 24);                    // store offsets: 24-bit multiply of the output row index n * P + p
}

template <int TT, int CW>   // compile-time tap count (0 = runtime T <= 8), chunk width in float4
__global__ __launch_bounds__(WGS_THREADS) void wgs_fwd_xcd_kernel(
    int ntasks, int bpt, int n, int k, int ldy, int T, int P, unsigned pmagic, int CV, int nchunk, int off, int offc,
    const float *__restrict__ Y, const int32_t *__restrict__ idx, const float *__restrict__ bias,
    int bias_bstride, float *__restrict__ out) {
    typedef wgs_vec_t vec_t;
    constexpr int MAXT = TT ? TT : 8, RL = WGS_THREADS / CW;
    const int seq = blockIdx.x >> 3;
    const int task = (seq / bpt) * 8 + (blockIdx.x & 7);
    if (task >= ntasks) return;
    const int b = task / nchunk, chunk = task - b * nchunk;
    const int cv = chunk * CW + (threadIdx.x % CW);
    if (cv >= CV) return;
    const int R = n * P;
    const int r0 = (seq % bpt) * (RL * WGS_XU) + threadIdx.x / CW;      // output rows r0 + u*RL of this sample
    const __amdgpu_buffer_rsrc_t rsY = wgs_rsrc(Y + (size_t)b * n * ldy, (unsigned)n * ldy * 4u);
    const __amdgpu_buffer_rsrc_t rsI = wgs_rsrc(idx + (size_t)b * n * k, (unsigned)n * k * 4u);
    const __amdgpu_buffer_rsrc_t rsO = wgs_rsrc(out + (size_t)b * R * CV * 4, (unsigned)R * CV * 16u);
    const unsigned cb = (unsigned)cv * 16u, ldy4 = (unsigned)ldy * 4u, C4 = (unsigned)CV * 16u;
    unsigned pt[WGS_XU], nb[WGS_XU][MAXT];
    bool live[WGS_XU];
#pragma unroll
    for (int u = 0; u < WGS_XU; ++u) {
        const int r = r0 + u * RL;
        live[u] = r < R;
        const unsigned rr = live[u] ? r : r0 < R ? r0 : 0;
        pt[u] = wgs_div(rr, P, pmagic);
        const unsigned io = (wgs_mul24(pt[u], k) + (rr - wgs_mul24(pt[u], P))) * 4u;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (TT || t < T) nb[u][t] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsI, (int)(io + 4u * t), 0, 0);
    }
    vec_t acc[WGS_XU], ctr[WGS_XU], tap[WGS_XU][MAXT];
    const vec_t zero = {0.f, 0.f, 0.f, 0.f};
    const vec_t bv = bias ? *reinterpret_cast<const vec_t *>(bias + (long long)b * bias_bstride + cv * 4) : zero;
#pragma unroll
    for (int u = 0; u < WGS_XU; ++u) {
        ctr[u] = offc >= 0 ? wgs_ld4(rsY, wgs_mul24(pt[u], ldy4) + cb, (unsigned)offc * 4u) : zero;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (TT || t < T) tap[u][t] = wgs_ld4(rsY, wgs_mul24(nb[u][t], ldy4) + cb, (unsigned)off * 4u + t * C4);
    }
#pragma unroll
    for (int u = 0; u < WGS_XU; ++u) {
        acc[u] = bv + ctr[u];
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (TT || t < T) acc[u] += tap[u][t];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wgs_u4_t, acc[u]), rsO,
                                               live[u] ? (int)(wgs_mul24(r0 + u * RL, C4) + cb) : (int)WGS_OOB, 0, 2 /* nt */);
    }
}

// The same gather-sum in the geometry of the BatchNorm reductions (bn_geom.h): a thread owns one float4 column group
// and walks its row lane of the block's rows, so it can also accumulate the per-column sum / sum of squares of what
// it writes.  part[blockIdx.y][c], [C + c] are exactly what cl_stats_kernel would produce for `out`: the BatchNorm
// that follows (inte_conv_hk.1) needs no statistics pass over the tensor.
template <int TT>   // compile-time tap count (inte_conv_hk: k/2+1 = 6 at k = 10), 0 = runtime T <= 8
__global__ __launch_bounds__(WGS_THREADS) void wgs_fwd_stats_kernel(
    long long R, int n, int k, int ldy, int T, int P, int C, int cgb, int rows_per_block, int off, int offc,
    const float *__restrict__ Y, const int32_t *__restrict__ idx, const float *__restrict__ bias, int bias_bstride,
    float *__restrict__ out, float *__restrict__ part) {
    __shared__ float4 red[2][WGS_THREADS];
    const int cgl = threadIdx.x % cgb, rlane = threadIdx.x / cgb, rl = WGS_THREADS / cgb;
    const int cgi = blockIdx.x * cgb + cgl;
    const bool cok = cgi * 4 < C;
    const int c = cgi * 4;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(R, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
    // A thread walks ~rows_per_block/rl rows one after the other, and a row is a two-level dependent chain (index ->
    // gathered row), so WGS_U rows are kept in flight at once: all their index loads, then all their row loads.
    constexpr int WGS_U = TT ? 4 : 2, WGS_MAXT = TT ? TT : 8;
    if (cok) {
        for (long long rb = r0 + rlane; rb < r1; rb += (long long)WGS_U * rl) {
            long long bn[WGS_U], bs[WGS_U];
            int nb[WGS_U][WGS_MAXT];
            bool live[WGS_U];
#pragma unroll
            for (int u = 0; u < WGS_U; ++u) {
                const long long r = rb + (long long)u * rl;
                live[u] = r < r1;
                const long long rr = live[u] ? r : rb;
                bn[u] = rr / P;
                bs[u] = bn[u] / n;
                const int32_t *I = idx + bn[u] * k + (int)(rr % P);
#pragma unroll
                for (int t = 0; t < WGS_MAXT; ++t)
                    if (TT || t < T) nb[u][t] = I[t];
            }
            float4 acc[WGS_U], ctr[WGS_U], tap[WGS_U][WGS_MAXT];
#pragma unroll
            for (int u = 0; u < WGS_U; ++u) {
                acc[u] = bias ? *reinterpret_cast<const float4 *>(bias + bs[u] * bias_bstride + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                ctr[u] = offc >= 0 ? *reinterpret_cast<const float4 *>(Y + bn[u] * ldy + offc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < WGS_MAXT; ++t)
                    if (TT || t < T) tap[u][t] = *reinterpret_cast<const float4 *>(Y + (bs[u] * n + nb[u][t]) * ldy + off + t * C + c);
            }
#pragma unroll
            for (int u = 0; u < WGS_U; ++u) {
                float4 a = acc[u];
                a.x += ctr[u].x; a.y += ctr[u].y; a.z += ctr[u].z; a.w += ctr[u].w;
#pragma unroll
                for (int t = 0; t < WGS_MAXT; ++t)
                    if (TT || t < T) { a.x += tap[u][t].x; a.y += tap[u][t].y; a.z += tap[u][t].z; a.w += tap[u][t].w; }
                if (live[u]) {
                    *reinterpret_cast<float4 *>(out + (rb + (long long)u * rl) * C + c) = a;
                    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                    q.x = __fmaf_rn(a.x, a.x, q.x); q.y = __fmaf_rn(a.y, a.y, q.y);
                    q.z = __fmaf_rn(a.z, a.z, q.z); q.w = __fmaf_rn(a.w, a.w, q.w);
                }
            }
        }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (rlane == 0 && cok) {
        for (int j = 1; j < rl; ++j) {
            const float4 a = red[0][j * cgb + cgl], b = red[1][j * cgb + cgl];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
        }
        float *Pp = part + (size_t)blockIdx.y * 2 * C;
        *reinterpret_cast<float4 *>(Pp + c) = s;
        *reinterpret_cast<float4 *>(Pp + C + c) = q;
    }
}

// wgs_fwd_stats in the task mapping of wgs_fwd_xcd_kernel: a block owns `rpb` output rows of one (sample, 64-channel
// chunk) task, keeps WGS_XU rows in flight per thread, and writes one partial row (row block `b*bpt + blk`) of the
// statistics; blocks past the tasks zero the partial rows the BatchNorm geometry has beyond b*bpt.
template <int TT, int CW>
__global__ __launch_bounds__(WGS_THREADS) void wgs_fwd_stats_xcd_kernel(
    int ntasks, int bpt, int rpb, int main_blocks, int gy_used, int n, int k, int ldy, int T, int P, unsigned pmagic, int CV,
    int nchunk, int off, int offc, const float *__restrict__ Y, const int32_t *__restrict__ idx, const float *__restrict__ bias,
    int bias_bstride, float *__restrict__ out, float *__restrict__ part) {
    typedef wgs_vec_t vec_t;
    constexpr int MAXT = TT ? TT : 8, RL = WGS_THREADS / CW;
    __shared__ vec_t red[2][WGS_THREADS];
    const int C = CV * 4;
    if ((int)blockIdx.x >= main_blocks) {                      // unused partial rows of the BatchNorm geometry: zeros
        float *Z = part + (size_t)(gy_used + (blockIdx.x - main_blocks)) * 2 * C;
        for (int i = threadIdx.x; i < 2 * C; i += WGS_THREADS) Z[i] = 0.f;
        return;
    }
    const int seq = blockIdx.x >> 3;
    const int task = (seq / bpt) * 8 + (blockIdx.x & 7);
    if (task >= ntasks) return;
    const int blk = seq % bpt;
    const int b = task / nchunk, chunk = task - b * nchunk;
    const int cvl = threadIdx.x % CW, rlane = threadIdx.x / CW;
    const int cv = chunk * CW + cvl;
    const bool cok = cv < CV;
    const int c = cv * 4, R = n * P;
    const vec_t zero = {0.f, 0.f, 0.f, 0.f};
    vec_t s = zero, q = zero;
    if (cok) {
        const __amdgpu_buffer_rsrc_t rsY = wgs_rsrc(Y + (size_t)b * n * ldy, (unsigned)n * ldy * 4u);
        const __amdgpu_buffer_rsrc_t rsI = wgs_rsrc(idx + (size_t)b * n * k, (unsigned)n * k * 4u);
        const __amdgpu_buffer_rsrc_t rsO = wgs_rsrc(out + (size_t)b * R * CV * 4, (unsigned)R * CV * 16u);
        const unsigned cb = (unsigned)cv * 16u, ldy4 = (unsigned)ldy * 4u, C4 = (unsigned)CV * 16u;
        const vec_t bv = bias ? *reinterpret_cast<const vec_t *>(bias + (long long)b * bias_bstride + c) : zero;
        const int rend = min(R, (blk + 1) * rpb);
        for (int r0 = blk * rpb + rlane; r0 < rend; r0 += RL * WGS_XU) {
            unsigned pt[WGS_XU], nb[WGS_XU][MAXT];
            bool live[WGS_XU];
#pragma unroll
            for (int u = 0; u < WGS_XU; ++u) {
                const int r = r0 + u * RL;
                live[u] = r < rend;
                const unsigned rr = live[u] ? r : r0;
                pt[u] = wgs_div(rr, P, pmagic);
                const unsigned io = (wgs_mul24(pt[u], k) + (rr - wgs_mul24(pt[u], P))) * 4u;
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
                    if (TT || t < T) nb[u][t] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsI, (int)(io + 4u * t), 0, 0);
            }
            vec_t ctr[WGS_XU], tap[WGS_XU][MAXT];
#pragma unroll
            for (int u = 0; u < WGS_XU; ++u) {
                ctr[u] = offc >= 0 ? wgs_ld4(rsY, wgs_mul24(pt[u], ldy4) + cb, (unsigned)offc * 4u) : zero;
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
                    if (TT || t < T) tap[u][t] = wgs_ld4(rsY, wgs_mul24(nb[u][t], ldy4) + cb, (unsigned)off * 4u + t * C4);
            }
#pragma unroll
            for (int u = 0; u < WGS_XU; ++u) {
                vec_t a = bv + ctr[u];
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
                    if (TT || t < T) a += tap[u][t];
                if (live[u]) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wgs_u4_t, a), rsO,
                                                           (int)(wgs_mul24(r0 + u * RL, C4) + cb), 0, 0);
                    s += a;
                    q.x = __fmaf_rn(a.x, a.x, q.x); q.y = __fmaf_rn(a.y, a.y, q.y);
                    q.z = __fmaf_rn(a.z, a.z, q.z); q.w = __fmaf_rn(a.w, a.w, q.w);
                }
            }
        }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (rlane == 0 && cok) {
        for (int j = 1; j < RL; ++j) {
            s += red[0][j * CW + cvl];
            q += red[1][j * CW + cvl];
        }
        float *Pp = part + (size_t)(b * bpt + blk) * 2 * C;
        *reinterpret_cast<vec_t *>(Pp + c) = s;
        *reinterpret_cast<vec_t *>(Pp + C + c) = q;
    }
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

// chunk width (float4 per row segment) of the task mapping; PDGN_WGS_CW / PDGN_WGS_BCW override for experiments
static int wgs_cw(const char *env, int dflt) {
    const char *v = getenv(env);
    const int w = v ? atoi(v) : dflt;
    return (w == 8 || w == 16 || w == 32 || w == 64) ? w : dflt;
}
#define WGS_DISPATCH_CW(cw, CALL)   \
    do {                            \
        if ((cw) == 8) { CALL(8); } else if ((cw) == 32) { CALL(32); } else if ((cw) == 64) { CALL(64); } else { CALL(16); } \
    } while (0)

static bool wgs_ok(int b, int n, int k, int ldy, int T, int P, int C, int off, int offc) {
    // T == 0: no taps, the centre column block (+ bias) alone -- out[b,n,p,c] = bias + Y[b,n,offc+c]
    return b >= 0 && n >= 1 && k >= 1 && T >= 0 && (T >= 1 || offc >= 0) && P >= 1 && C >= 1 && T + P - 1 <= k && off >= 0 &&
           off + T * C <= ldy && (offc < 0 || offc + C <= ldy);
}

extern "C" int pdgn_window_gather_sum(int b, int n, int k, int ldy, int T, int P, int C, int off, int offc,
                                      const float *Y, const int32_t *idx, const float *bias, int bias_bstride,
                                      float *out,
                                      pdgn_stream_t stream) {
    if (!wgs_ok(b, n, k, ldy, T, P, C, off, offc)) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = (C % 4 == 0) && (ldy % 4 == 0) && (off % 4 == 0) && (offc < 0 || offc % 4 == 0) &&
                    (!bias || (bias_bstride % 4 == 0 && ((size_t)bias & 15) == 0));
    static const int xcd = getenv("PDGN_WGS_XCD") ? atoi(getenv("PDGN_WGS_XCD")) : 1;
    if (v4 && xcd && T <= 8 && (long long)n * P * (C / 4) >= 65536 && wgs_slabs_ok(n, k, ldy, P, C)) {   // enough re-read volume for the L2 mapping to matter
        static const int cw = wgs_cw("PDGN_WGS_CW", 32);
        const int CV = C / 4, nchunk = cdiv(CV, cw), ntasks = b * nchunk;
        const int bpt = cdiv((long long)n * P, WGS_THREADS / cw * WGS_XU);
        const long long blocks = (long long)cdiv(ntasks, 8) * 8 * bpt;
        if (blocks > 0x7fffffffLL) return PDGN_ERR_INVALID;
#define WGS_CALL(W)                                                                                                        \
    if (T == 6)                                                                                                            \
        hipLaunchKernelGGL((wgs_fwd_xcd_kernel<6, W>), dim3((unsigned)blocks), dim3(WGS_THREADS), 0, s, ntasks, bpt, n, k, \
                           ldy, T, P, wgs_magic(P), CV, nchunk, off, offc, Y, idx, bias, bias_bstride, out);               \
    else                                                                                                                   \
        hipLaunchKernelGGL((wgs_fwd_xcd_kernel<0, W>), dim3((unsigned)blocks), dim3(WGS_THREADS), 0, s, ntasks, bpt, n, k, \
                           ldy, T, P, wgs_magic(P), CV, nchunk, off, offc, Y, idx, bias, bias_bstride, out)
        WGS_DISPATCH_CW(cw, WGS_CALL);
#undef WGS_CALL
    } else if (v4) {
        long long total = (long long)b * n * P * (C / 4);
        hipLaunchKernelGGL(wgs_fwd_kernel<4>, dim3(cdiv(total, WGS_THREADS)), dim3(WGS_THREADS), 0, s, total,
                           n, k, ldy, T, P, C / 4, off, offc, Y, idx, bias, bias_bstride, out);
    } else {
        long long total = (long long)b * n * P * C;
        hipLaunchKernelGGL(wgs_fwd_kernel<1>, dim3(cdiv(total, WGS_THREADS)), dim3(WGS_THREADS), 0, s, total,
                           n, k, ldy, T, P, C, off, offc, Y, idx, bias, bias_bstride, out);
    }
    return pdgn_launch_status();
}

// pdgn_window_gather_sum + the BatchNorm partial statistics of `out` viewed as (b*n*P, C) rows: `scratch` (>=
// pdgn_bn_scratch_floats(b*n*P, C) floats) receives what pdgn_bn_stats' first stage would compute; finish with
// pdgn_bn_stats_from_partials.  Needs the float4 path (C, ldy, off, offc multiples of 4), else -1.
extern "C" int pdgn_window_gather_sum_stats(int b, int n, int k, int ldy, int T, int P, int C, int off, int offc,
                                            const float *Y, const int32_t *idx, const float *bias, int bias_bstride,
                                            float *out, float *scratch, pdgn_stream_t stream) {
    if (!wgs_ok(b, n, k, ldy, T, P, C, off, offc) || b < 1 || T > 8) return PDGN_ERR_INVALID;
    if ((C % 4) || (ldy % 4) || (off % 4) || (offc >= 0 && offc % 4) || (bias && bias_bstride % 4)) return PDGN_ERR_INVALID;
    const long long R = (long long)b * n * P;
    int cgb, gx, gy, rpb;
    cl_geometry(R, C, &cgb, &gx, &gy, &rpb);
    static const int xcd = getenv("PDGN_WGS_XCD") ? atoi(getenv("PDGN_WGS_XCD")) : 1;
    if (xcd && (long long)n * P * (C / 4) >= 65536 && gy >= b && wgs_slabs_ok(n, k, ldy, P, C)) {
        // partial rows: one per (sample, row block); the BatchNorm geometry's gy bounds them, the rest are zeroed
        static const int cw = wgs_cw("PDGN_WGS_SCW", 16);
        const int RLU = WGS_THREADS / cw * WGS_XU;
        int bpt = gy / b;
        int rows = cdiv((long long)n * P, bpt);
        rows = cdiv(rows, RLU) * RLU;
        bpt = cdiv((long long)n * P, rows);
        const int CV = C / 4, nchunk = cdiv(CV, cw), ntasks = b * nchunk;
        const long long main_blocks = (long long)cdiv(ntasks, 8) * 8 * bpt, blocks = main_blocks + (gy - b * bpt);
        if (blocks <= 0x7fffffffLL && rows <= 65536) {
#define WGS_CALL(W)                                                                                                             \
    if (T == 6)                                                                                                                 \
        hipLaunchKernelGGL((wgs_fwd_stats_xcd_kernel<6, W>), dim3((unsigned)blocks), dim3(WGS_THREADS), 0, (hipStream_t)stream, \
                           ntasks, bpt, rows, (int)main_blocks, b * bpt, n, k, ldy, T, P, wgs_magic(P), CV, nchunk, off, offc,  \
                           Y, idx, bias, bias_bstride, out, scratch);                                                           \
    else                                                                                                                        \
        hipLaunchKernelGGL((wgs_fwd_stats_xcd_kernel<0, W>), dim3((unsigned)blocks), dim3(WGS_THREADS), 0, (hipStream_t)stream, \
                           ntasks, bpt, rows, (int)main_blocks, b * bpt, n, k, ldy, T, P, wgs_magic(P), CV, nchunk, off, offc,  \
                           Y, idx, bias, bias_bstride, out, scratch)
            WGS_DISPATCH_CW(cw, WGS_CALL);
#undef WGS_CALL
            return pdgn_launch_status();
        }
    }
    if (T == 6)
        hipLaunchKernelGGL(wgs_fwd_stats_kernel<6>, dim3(gx, gy), dim3(WGS_THREADS), 0, (hipStream_t)stream, R, n, k, ldy, T, P,
                           C, cgb, rpb, off, offc, Y, idx, bias, bias_bstride, out, scratch);
    else
        hipLaunchKernelGGL(wgs_fwd_stats_kernel<0>, dim3(gx, gy), dim3(WGS_THREADS), 0, (hipStream_t)stream, R, n, k, ldy, T, P,
                           C, cgb, rpb, off, offc, Y, idx, bias, bias_bstride, out, scratch);
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

// ---------------------------------------------------------------------------- atomic-free adjoint
// Transposed kNN graph (CSR over source points): for every point j the list of edges (n, s) with
// idx[b,n,s] == j, packed as n*32 + s.  With it the adjoint of the gather-sum becomes a gather
// itself -- every dY element is written exactly once (no zero-fill pass, no float atomics):
//   dY[b,j,off+t*C+c] = sum_{(n,s) in in(j), 0 <= s-t < P} dout[b,n,s-t,c]
//   dY[b,j,offc+c]    = sum_p dout[b,j,p,c]
__global__ __launch_bounds__(WGS_THREADS) void csr_count_kernel(long long total, int n, int k,
                                                                const int32_t *__restrict__ idx,
                                                                int32_t *__restrict__ cnt) {
    long long e = (long long)blockIdx.x * WGS_THREADS + threadIdx.x;
    if (e >= total) return;
    const long long b = e / ((long long)n * k);
    atomicAdd(&cnt[b * n + idx[e]], 1);
}

// one workgroup per batch: exclusive scan of cnt (n) -> rowptr (n+1); cursor <- rowptr
__global__ __launch_bounds__(1024) void csr_scan_kernel(int n, const int32_t *__restrict__ cnt,
                                                        int32_t *__restrict__ rowptr, int32_t *__restrict__ cursor) {
    __shared__ int part[1024];
    const int bs = blockIdx.x, tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int32_t *C = cnt + (size_t)bs * n;
    int s = 0;
    for (int i = tid * per; i < min(n, (tid + 1) * per); ++i) s += C[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;                                  // exclusive prefix of this thread's chunk
    int32_t *R = rowptr + (size_t)bs * (n + 1), *U = cursor + (size_t)bs * n;
    for (int i = tid * per; i < min(n, (tid + 1) * per); ++i) {
        R[i] = run; U[i] = run;
        run += C[i];
    }
    if (tid == 1023) R[n] = part[1023];
}

__global__ __launch_bounds__(WGS_THREADS) void csr_fill_kernel(long long total, int n, int k,
                                                               const int32_t *__restrict__ idx,
                                                               int32_t *__restrict__ cursor,
                                                               int32_t *__restrict__ edges) {
    long long e = (long long)blockIdx.x * WGS_THREADS + threadIdx.x;
    if (e >= total) return;
    const long long nk = (long long)n * k;
    const long long b = e / nk;
    const int r = (int)(e - b * nk);                          // n_local * k + s
    const int pos = atomicAdd(&cursor[b * n + idx[e]], 1);
    edges[b * nk + pos] = (r / k) * 32 + (r % k);
}

// max_out (the adjoint kernels; may be NULL): uint32[b n], the maximum of |dY| (bit pattern) over each ROW (b, j) of dY -- what the
// two-part contractions that take dY as their first operand (the per-point product's input gradient, gemm_x3.hip: one power-of-two
// scale per row) would otherwise scan its 1.8 GB for.  Atomic max (order-independent); the launcher of a dY's FIRST spec
// zero-fills the array.
__device__ __forceinline__ unsigned wgs_absmax4(const float __attribute__((ext_vector_type(4))) v) {
    return max(max(__float_as_uint(v[0]) & 0x7fffffffu, __float_as_uint(v[1]) & 0x7fffffffu),
               max(__float_as_uint(v[2]) & 0x7fffffffu, __float_as_uint(v[3]) & 0x7fffffffu));
}

// (the small shapes' kernel -- the 16-channel branches, the first levels' specs.  Round 6: EL = 8 consecutive lanes share one
// (source point, tap, float4 column) and split its in-edges (or, for the centre columns, its P windows) between them: one thread
// walking all in-edges of a hub was a chain of up to 60 dependent loads -- 55 us per launch for 23 MB at stage 4, 0.44 ms per
// iteration on the issuing stream; the eight partial sums meet in three xor-shuffles, a fixed order)
template <int EL>
__global__ __launch_bounds__(WGS_THREADS) void wgs_bwd_csr_kernel(
    long long total, int n, int k, int ldy, int T, int P, int CV, int off, int offc,
    const float *__restrict__ dout, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ edges,
    float *__restrict__ dY, unsigned *__restrict__ max_out) {
    typedef float vec_t __attribute__((ext_vector_type(4)));
    static_assert(64 % EL == 0, "the lanes of an item lie in one wave");
    const long long g = (long long)blockIdx.x * WGS_THREADS + threadIdx.x;
    const long long e = g / EL;
    const int el = (int)(g % EL);
    const bool valid = e < total;
    vec_t acc = {0.f, 0.f, 0.f, 0.f};
    const int TT = offc >= 0 ? T + 1 : T;
    const int cv = (int)(e % CV);
    const long long r = e / CV;
    const int t = (int)(r % TT);
    const long long bj = r / TT;                              // b * n + j
    const long long b = bj / n;
    const int j = (int)(bj - b * n);
    const int C = CV * 4, c = cv * 4;
    if (valid) {
        if (t == T) {                                         // centre columns
            const float *src = dout + bj * P * C + c;
            for (int p = el; p < P; p += EL) acc += *reinterpret_cast<const vec_t *>(src + (size_t)p * C);
        } else {
            const int32_t *R = rowptr + b * (n + 1);
            const int32_t *E = edges + b * (long long)n * k;
            const int e1 = R[j + 1];
            for (int q = R[j] + el; q < e1; q += EL) {
                const int rec = E[q];
                const int p = (rec & 31) - t;
                if (p >= 0 && p < P)
                    acc += *reinterpret_cast<const vec_t *>(dout + ((b * n + (rec >> 5)) * P + p) * C + c);
            }
        }
    }
#pragma unroll
    for (int o = 1; o < EL; o <<= 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], o);
    if (!valid || el != 0) return;
    *reinterpret_cast<vec_t *>(dY + bj * ldy + (t == T ? offc : off + t * C) + c) = acc;
    if (max_out) atomicMax(max_out + bj, wgs_absmax4(acc));       // (one atomic per 16 B written)
}

// The adjoint in the task mapping of wgs_fwd_xcd_kernel.  A WAVE owns one source point j of a (sample, 64-channel
// chunk) task: lane = (float4 column, edge lane); the four edge lanes of a column split j's in-edges between them (the
// in-degree of a kNN graph is skewed: a hub no longer holds up three neighbours' lanes), an in-edge (n', s) is decoded
// once and feeds dout[n', s-t] to accumulator t for every valid t -- up to 2P independent loads in flight -- and the
// four partial sums meet in two cross-lane exchanges.  The task's dout chunk (n*P rows x 256 B = 1.3 MB at stage 4, each
// row wanted by ~T/2 taps of k sources) stays in one XCD's L2.
template <int TT, int CW>
__global__ __launch_bounds__(WGS_THREADS) void wgs_bwd_csr_xcd_kernel(
    int ntasks, int bpt, int n, int k, int ldy, int T, int P, int CV, int nchunk, int off, int offc,
    const float *__restrict__ dout, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ edges,
    float *__restrict__ dY, unsigned *__restrict__ max_out) {
    typedef wgs_vec_t vec_t;
    constexpr int MAXT = TT ? TT : 8, JB = WGS_THREADS / 64, WGS_EL = 64 / CW;   // source points per block; edge lanes per column
    const int seq = blockIdx.x >> 3;
    const int task = (seq / bpt) * 8 + (blockIdx.x & 7);
    if (task >= ntasks) return;
    const int b = task / nchunk, chunk = task - b * nchunk;
    const int lane = threadIdx.x & 63, cvl = lane % CW, el = lane / CW;
    const int cv = chunk * CW + cvl;
    const int j = (seq % bpt) * JB + (threadIdx.x >> 6);
    if (j >= n) return;                                        // wave-uniform
    const bool cok = cv < CV;
    // one sample's slabs through buffer descriptors, 32-bit offsets (see wgs_fwd_xcd_kernel)
    const unsigned C4 = (unsigned)CV * 16u, PC4 = (unsigned)P * C4, cb = (unsigned)(cok ? cv : 0) * 16u, ldy4 = (unsigned)ldy * 4u;
    const __amdgpu_buffer_rsrc_t rsD = wgs_rsrc(dout + (size_t)b * n * P * CV * 4, (unsigned)n * PC4);
    const __amdgpu_buffer_rsrc_t rsE = wgs_rsrc(edges + (size_t)b * n * k, (unsigned)n * k * 4u);
    const __amdgpu_buffer_rsrc_t rsY = wgs_rsrc(dY + (size_t)b * n * ldy, (unsigned)n * ldy4);
    const vec_t zero = {0.f, 0.f, 0.f, 0.f};
    vec_t acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = zero;
    const int e0 = rowptr[(long long)b * (n + 1) + j], e1 = rowptr[(long long)b * (n + 1) + j + 1];
    for (int q = e0 + el; q < e1; q += 2 * WGS_EL) {           // this lane's edges, two at a time
        const bool two = q + WGS_EL < e1;
        const unsigned ra = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsE, q * 4, 0, 0);
        const unsigned rb = two ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsE, (q + WGS_EL) * 4, 0, 0) : 0u;
        const unsigned oa = wgs_mul24(ra >> 5, PC4) + cb, ob = wgs_mul24(rb >> 5, PC4) + cb;
        const int sa = ra & 31, sb = two ? (int)(rb & 31) : -64;
        vec_t va[MAXT], vb[MAXT];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int pa = sa - t, pb = sb - t;                // out-of-window taps issue no load (exec-masked)
            va[t] = ((TT || t < T) && pa >= 0 && pa < P) ? wgs_ld4(rsD, oa + wgs_mul24(pa, C4), 0) : zero;
            vb[t] = ((TT || t < T) && pb >= 0 && pb < P) ? wgs_ld4(rsD, ob + wgs_mul24(pb, C4), 0) : zero;
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t) acc[t] += va[t] + vb[t];
    }
    vec_t ctr = zero;
    if (offc >= 0)                                             // centre columns: the point's own P windows, split too
        for (int p = el; p < P; p += WGS_EL) ctr += wgs_ld4(rsD, wgs_mul24(j, PC4) + wgs_mul24(p, C4) + cb, 0);
    // sum over the four edge lanes (lanes l, l+16, l+32, l+48): afterwards every lane holds the totals
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
        if (TT || t < T) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[t][i];
#pragma unroll
                for (int m = CW; m < 64; m <<= 1) v += __shfl_xor(v, m);
                acc[t][i] = v;
            }
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = ctr[i];
#pragma unroll
        for (int m = CW; m < 64; m <<= 1) v += __shfl_xor(v, m);
        ctr[i] = v;
    }
    if (cok) {
        const unsigned oj = wgs_mul24(j, ldy4) + cb;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)                         // the stores are dealt out over the edge lanes as well
            if ((TT || t < T) && (t % WGS_EL) == el)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wgs_u4_t, acc[t]), rsY, (int)oj, (int)((unsigned)off * 4u + t * C4), 0);
        if (offc >= 0 && el == WGS_EL - 1)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wgs_u4_t, ctr), rsY, (int)oj, (int)((unsigned)offc * 4u), 0);
    }
    if (max_out) {                                             // (wave-uniform: j < n) the totals every lane holds -> one atomic per wave
        vec_t m = zero;
        if (cok) {
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
                if (TT || t < T)
#pragma unroll
                    for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], fabsf(acc[t][i]));
            if (offc >= 0)
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], fabsf(ctr[i]));
        }
        unsigned mm = wgs_absmax4(m);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mm = max(mm, (unsigned)__shfl_xor((int)mm, o));
        if (lane == 0) atomicMax(max_out + (size_t)b * n + j, mm);
    }
}

extern "C" int pdgn_knn_graph_transpose(int b, int n, int k, const int32_t *idx, int32_t *rowptr,
                                        int32_t *edges, int32_t *scratch, pdgn_stream_t stream) {
    if (b < 0 || n < 1 || k < 1 || k > 31 || (long long)n * 32 > 0x7fffffffLL) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int32_t *cnt = scratch, *cursor = scratch + (size_t)b * n;      // scratch: 2*b*n ints
    hipError_t e = hipMemsetAsync(cnt, 0, (size_t)b * n * sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    const long long total = (long long)b * n * k;
    hipLaunchKernelGGL(csr_count_kernel, dim3(cdiv(total, WGS_THREADS)), dim3(WGS_THREADS), 0, s, total, n, k, idx, cnt);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(b), dim3(1024), 0, s, n, cnt, rowptr, cursor);
    hipLaunchKernelGGL(csr_fill_kernel, dim3(cdiv(total, WGS_THREADS)), dim3(WGS_THREADS), 0, s, total, n, k, idx,
                       cursor, edges);
    return pdgn_launch_status();
}

extern "C" int pdgn_window_gather_sum_backward_csr(int b, int n, int k, int ldy, int T, int P, int C, int off,
                                                   int offc, const float *dout, const int32_t *rowptr,
                                                   const int32_t *edges, float *dY, unsigned *max_out, int max_init,
                                                   pdgn_stream_t stream) {
    if (!wgs_ok(b, n, k, ldy, T, P, C, off, offc) || C % 4 || ldy % 4 || off % 4 || (offc >= 0 && offc % 4) || k > 31)
        return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    if (max_out && max_init && hipMemsetAsync(max_out, 0, (size_t)b * n * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return pdgn_launch_status();
    static const int xcd = getenv("PDGN_WGS_XCD") ? atoi(getenv("PDGN_WGS_XCD")) : 1;
    if (xcd && (T <= 8 || T == 10) && (long long)n * T * (C / 4) >= 65536 && wgs_slabs_ok(n, k, ldy, P, C)) {
        static const int cw = wgs_cw("PDGN_WGS_BCW", 64);
        const int CV = C / 4, nchunk = cdiv(CV, cw), ntasks = b * nchunk, bpt = cdiv(n, WGS_THREADS / 64);
        const long long blocks = (long long)cdiv(ntasks, 8) * 8 * bpt;
        if (blocks <= 0x7fffffffLL) {
#define WGS_CALL(W)                                                                                                           \
    if (T == 6)                                                                                                               \
        hipLaunchKernelGGL((wgs_bwd_csr_xcd_kernel<6, W>), dim3((unsigned)blocks), dim3(WGS_THREADS), 0, (hipStream_t)stream, \
                           ntasks, bpt, n, k, ldy, T, P, CV, nchunk, off, offc, dout, rowptr, edges, dY, max_out);                     \
    else if (T == 10)                                                                                                         \
        hipLaunchKernelGGL((wgs_bwd_csr_xcd_kernel<10, W>), dim3((unsigned)blocks), dim3(WGS_THREADS), 0, (hipStream_t)stream,\
                           ntasks, bpt, n, k, ldy, T, P, CV, nchunk, off, offc, dout, rowptr, edges, dY, max_out);                     \
    else                                                                                                                      \
        hipLaunchKernelGGL((wgs_bwd_csr_xcd_kernel<0, W>), dim3((unsigned)blocks), dim3(WGS_THREADS), 0, (hipStream_t)stream, \
                           ntasks, bpt, n, k, ldy, T, P, CV, nchunk, off, offc, dout, rowptr, edges, dY, max_out)
            WGS_DISPATCH_CW(cw, WGS_CALL);
#undef WGS_CALL
            return pdgn_launch_status();
        }
    }
    const int TT = offc >= 0 ? T + 1 : T;
    const long long total = (long long)b * n * TT * (C / 4);
    hipLaunchKernelGGL(wgs_bwd_csr_kernel<8>, dim3(cdiv(total * 8, WGS_THREADS)), dim3(WGS_THREADS), 0, (hipStream_t)stream,
                       total, n, k, ldy, T, P, C / 4, off, offc, dout, rowptr, edges, dY, max_out);
    return pdgn_launch_status();
}
