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
                                      const float *Y, const int32_t *idx, const float *bias, int bias_bstride,
                                      float *out,
                                      pdgn_stream_t stream) {
    if (!wgs_ok(b, n, k, ldy, T, P, C, off, offc)) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = (C % 4 == 0) && (ldy % 4 == 0) && (off % 4 == 0) && (offc < 0 || offc % 4 == 0) &&
                    (!bias || (bias_bstride % 4 == 0 && ((size_t)bias & 15) == 0));
    if (v4) {
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

__global__ __launch_bounds__(WGS_THREADS) void wgs_bwd_csr_kernel(
    long long total, int n, int k, int ldy, int T, int P, int CV, int off, int offc,
    const float *__restrict__ dout, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ edges,
    float *__restrict__ dY) {
    typedef float vec_t __attribute__((ext_vector_type(4)));
    long long e = (long long)blockIdx.x * WGS_THREADS + threadIdx.x;
    if (e >= total) return;
    const int TT = offc >= 0 ? T + 1 : T;
    const int cv = (int)(e % CV);
    long long r = e / CV;
    const int t = (int)(r % TT);
    const long long bj = r / TT;                              // b * n + j
    const long long b = bj / n;
    const int j = (int)(bj - b * n);
    const int C = CV * 4, c = cv * 4;
    vec_t acc = {0.f, 0.f, 0.f, 0.f};
    if (t == T) {                                             // centre columns
        const float *src = dout + bj * P * C + c;
        for (int p = 0; p < P; ++p) acc += *reinterpret_cast<const vec_t *>(src + (size_t)p * C);
        *reinterpret_cast<vec_t *>(dY + bj * ldy + offc + c) = acc;
        return;
    }
    const int32_t *R = rowptr + b * (n + 1);
    const int32_t *E = edges + b * (long long)n * k;
    const int e1 = R[j + 1];
    for (int q = R[j]; q < e1; ++q) {
        const int rec = E[q];
        const int p = (rec & 31) - t;
        if (p >= 0 && p < P)
            acc += *reinterpret_cast<const vec_t *>(dout + ((b * n + (rec >> 5)) * P + p) * C + c);
    }
    *reinterpret_cast<vec_t *>(dY + bj * ldy + off + t * C + c) = acc;
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
                                                   const int32_t *edges, float *dY, pdgn_stream_t stream) {
    if (!wgs_ok(b, n, k, ldy, T, P, C, off, offc) || C % 4 || ldy % 4 || off % 4 || (offc >= 0 && offc % 4) || k > 31)
        return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    const int TT = offc >= 0 ? T + 1 : T;
    const long long total = (long long)b * n * TT * (C / 4);
    hipLaunchKernelGGL(wgs_bwd_csr_kernel, dim3(cdiv(total, WGS_THREADS)), dim3(WGS_THREADS), 0, (hipStream_t)stream,
                       total, n, k, ldy, T, P, C / 4, off, offc, dout, rowptr, edges, dY);
    return pdgn_launch_status();
}
