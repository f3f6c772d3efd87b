// feat_knn.hip -- feature-space kNN graph of the point-deconvolution blocks (pdgn_feature_knn).
//
// Semantics: models/PDGNet_v2.py:447-458 / :488-502 of the reference --
//   dist[i,j] = (-2 * <x_i, x_j> + |x_i|^2) + |x_j|^2 ;  idx = argsort(dist[i,:])[1 : k+1]
// i.e. the top-(k+1) of every row with rank 0 dropped (rank 0 is *presumed* self, not checked).
// Ties are ordered by index (the total order (dist, j)).
//
// The reference materialises the (B,N,N) matrix with bmm and fully sorts every row.  Here one
// workgroup owns 32 query points: the Gram tile is produced on the matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chains over the feature axis, candidates on the MFMA
// rows, queries on the lanes), written once to LDS as dist[query][candidate], and each query is
// then selected by a whole wave with the threshold/compaction/bitonic scheme of wave_select.h.
// Candidates are walked in chunks of 256 (33 KB of LDS per chunk, two workgroups per CU so one
// workgroup's MFMA phase overlaps the other's selection phase).  Nothing of size N^2 ever
// reaches HBM.
#include "common.h"
#include "wave_select.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FK_THREADS 256
#define FK_WAVES 4
#define FK_QB 32                 // queries per workgroup
#define FK_MAX_K 31              // k+1 <= 32 lanes of running list

// |x_i|^2 over the channel axis: x (b,f,n) -> sq (b,n)
__global__ __launch_bounds__(256) void sqnorm_kernel(int f, int n, const float *__restrict__ x,
                                                     float *__restrict__ sq) {
    const int bs = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *X = x + (size_t)bs * f * n;
    float s = 0.f;
    for (int c = 0; c < f; ++c) s = __fmaf_rn(X[(size_t)c * n + i], X[(size_t)c * n + i], s);
    sq[(size_t)bs * n + i] = s;
}

template <int FH, int NC>
__global__ __launch_bounds__(FK_THREADS) void feat_knn_kernel(
    int f, int n, int k, const float *__restrict__ x, const float *__restrict__ sq,
    int32_t *__restrict__ idx) {
    constexpr int NCP = NC + 4;
    __shared__ float dist[FK_QB][NCP];
    __shared__ float bqs[FH][PDGN_WAVE];         // B operand (queries), lane-indexed
    __shared__ float sqc[NC];
    __shared__ DI queue[FK_WAVES][WSEL_QCAP];

    const int bs = blockIdx.y;
    const int q0 = blockIdx.x * FK_QB;
    const int lane = lane_id();
    const int wave = threadIdx.x / PDGN_WAVE;
    const int col = lane & 31, half = lane >> 5;
    const float *X = x + (size_t)bs * f * n;
    const float *SQ = sq + (size_t)bs * n;
    const int K = k + 1;

    // B operand: bqs[s][lane] = channel 2s+half of query q0+col (zero-padded); every wave's
    // MFMA reads it back with one conflict-free ds_read_b32 per step.
    const int qcol = q0 + col;
    for (int s = wave; s < FH; s += FK_WAVES) {
        int c = 2 * s + half;
        bqs[s][lane] = (c < f && qcol < n) ? X[(size_t)c * n + qcol] : 0.f;
    }
    const float sq_q = qcol < n ? SQ[qcol] : 0.f;

    float rd[FK_QB / FK_WAVES];
    int ri[FK_QB / FK_WAVES];
#pragma unroll
    for (int t = 0; t < FK_QB / FK_WAVES; ++t) { rd[t] = INFINITY; ri[t] = 0x7fffffff; }

    for (int t0 = 0; t0 < n; t0 += NC) {
        const int tn = min(NC, n - t0);
        __syncthreads();                         // previous chunk's selection finished (and bqs ready)
        for (int c = threadIdx.x; c < tn; c += FK_THREADS) sqc[c] = SQ[t0 + c];
        __syncthreads();
        // ---- Gram tiles on the matrix cores
        for (int rt = wave; rt * 32 < tn; rt += FK_WAVES) {
            const int r = t0 + rt * 32 + col;    // candidate row this lane feeds to the A operand
            const float *Ar = X + (size_t)half * n + min(r, n - 1);
            const bool rok = r < n;
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
            for (int s = 0; s < FH; ++s) {
                int c = 2 * s + half;
                float a = (rok && c < f) ? Ar[(size_t)(2 * s) * n] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bqs[s][lane], acc, 0, 0, 0);
            }
            // D[row = (reg&3) + 8*(reg>>2) + 4*half][col]: candidate rows, query column
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = rt * 32 + 8 * g + 4 * half;      // local candidate index of reg 4g
                float4 v;
                float *vp = reinterpret_cast<float *>(&v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = __fmaf_rn(-2.0f, acc[4 * g + e], sq_q) + sqc[min(lc + e, NC - 1)];
                    vp[e] = (lc + e < tn) ? d : INFINITY;
                }
                *reinterpret_cast<float4 *>(&dist[col][lc]) = v;
            }
        }
        __syncthreads();
        // ---- selection: wave w owns queries 8w .. 8w+7
#pragma unroll
        for (int t = 0; t < FK_QB / FK_WAVES; ++t) {
            const int ql = wave * (FK_QB / FK_WAVES) + t;
            if (q0 + ql < n) {
                const float *row = dist[ql];
                wave_topk_scan([&](int c) { return row[c]; }, tn, t0, queue[wave], K, rd[t], ri[t], lane);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < FK_QB / FK_WAVES; ++t) {
        const int qq = q0 + wave * (FK_QB / FK_WAVES) + t;
        if (qq < n && lane >= 1 && lane <= k)
            idx[((size_t)bs * n + qq) * k + lane - 1] = rd[t] < INFINITY ? ri[t] : 0;   // rank 0 dropped (:458, :501)
    }
}

template <int FH>
static int launch_fk(int b, int f, int n, int k, const float *x, const float *sq, int32_t *idx,
                     hipStream_t s) {
    dim3 grid(cdiv(n, FK_QB), b);
    hipLaunchKernelGGL((feat_knn_kernel<FH, 256>), grid, dim3(FK_THREADS), 0, s, f, n, k, x, sq, idx);
    return pdgn_launch_status();
}

extern "C" int pdgn_feature_knn(int b, int f, int n, int k, const float *x, float *sqnorm,
                                int32_t *idx, pdgn_stream_t stream) {
    if (b < 0 || f < 1 || f > 256 || n < 1 || k < 1 || k > FK_MAX_K || n < k + 1 || b > 65535)
        return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, s, f, n, x, sqnorm);
    int rc = pdgn_launch_status();
    if (rc) return rc;
    const int fh = (f + 1) / 2;
    if (fh <= 16) return launch_fk<16>(b, f, n, k, x, sqnorm, idx, s);
    if (fh <= 32) return launch_fk<32>(b, f, n, k, x, sqnorm, idx, s);
    if (fh <= 64) return launch_fk<64>(b, f, n, k, x, sqnorm, idx, s);
    return launch_fk<128>(b, f, n, k, x, sqnorm, idx, s);
}
