// feat_knn.hip -- feature-space kNN graph of the point-deconvolution blocks (pdgn_feature_knn).
//
// Semantics: models/PDGNet_v2.py:447-458 / :488-502 of the reference --
//   dist[i,j] = (-2 * <x_i, x_j> + |x_i|^2) + |x_j|^2 ;  idx = argsort(dist[i,:])[1 : k+1]
// i.e. the top-(k+1) of every row with rank 0 dropped (rank 0 is *presumed* self, not checked).
// Ties are ordered by index (the total order (dist, j)).
//
// The reference materialises the (B,N,N) matrix with bmm and fully sorts every row.  Here one
// workgroup owns 32 query points and walks the candidates in chunks of 512:
//   * Gram tile on the matrix cores (v_mfma_f32_32x32x2_f32 = exact fp32 fma chains over the
//     feature axis): queries on the MFMA columns (B operand, staged once in LDS), candidates on
//     the rows.  Each wave owns a 128-candidate strip and feeds it with ONE 16-byte load per lane
//     per channel pair: lane (h,a,b) reads x[2s+h][strip + 32a + 4b .. +3] and element j of that
//     float4 is the A operand of accumulator j, i.e. accumulator j holds candidates
//     strip + 32a + 4b + j (the MFMA does not care which candidate sits on which row) -- four
//     MFMAs per global load, fully coalesced 512-byte segments.
//   * the strip is written to LDS as dist[query][candidate]; every query is then selected by a
//     whole wave: per-lane minima -> exact K-th smallest by ballot bisection (tau) -> survivors
//     d <= tau compacted into a per-query LDS queue; one (distance, index) bitonic sort per query
//     at the end (wave_select.h).
// Nothing of size N^2 ever reaches HBM.
#include "common.h"
#ifdef FK_TIMING
__device__ unsigned long long fk_dbg2[16];
#define WSEL_T(n) do { if (blockIdx.x == 0 && threadIdx.x == 256) fk_dbg2[n] = clock64(); } while (0)
extern "C" int fk_dbg2_copy(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fk_dbg2), sizeof(fk_dbg2)); }
#endif
#include "wave_select.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FK_THREADS 256
#define FK_WAVES 4
#define FK_QB 32                 // queries per workgroup
#define FK_QPW (FK_QB / FK_WAVES)
#define FK_NC 512                // candidates per chunk = FK_WAVES strips of 128
#define FK_MAX_K 31              // k+1 <= 32 lanes of running list

// |x_i|^2 over the channel axis: x (b,f,n) -> sq (b,n)
__global__ __launch_bounds__(256) void sqnorm_kernel(int f, int n, const float *__restrict__ x,
                                                     float *__restrict__ sq) {
    const int bs = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *X = x + (size_t)bs * f * n;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= f; c += 8) {                             // eight loads in flight; the fma chain keeps its channel order
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = X[(size_t)(c + j) * n + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) s = __fmaf_rn(v[j], v[j], s);
    }
    for (; c < f; ++c) s = __fmaf_rn(X[(size_t)c * n + i], X[(size_t)c * n + i], s);
    sq[(size_t)bs * n + i] = s;
}

template <int FH>
__global__ __launch_bounds__(FK_THREADS) void feat_knn_kernel(
    int f, int n, int k, const float *__restrict__ x, const float *__restrict__ sq,
    int32_t *__restrict__ idx) {
    constexpr int NCP = FK_NC + 4;
    __shared__ float dist[FK_QB][NCP];
    __shared__ float bqs[FH][PDGN_WAVE];         // B operand (queries), lane-indexed
    __shared__ float sqc[FK_NC];
    __shared__ DI queue[FK_QB][WSEL_PQCAP];

    const int bs = blockIdx.y;
    const int q0 = blockIdx.x * FK_QB;
    const int lane = lane_id();
    const int wave = threadIdx.x / PDGN_WAVE;
    const int col = lane & 31, half = lane >> 5;
    const float *X = x + (size_t)bs * f * n;
    const float *SQ = sq + (size_t)bs * n;
    const int K = k + 1;
    const bool vec_ok = (n & 3) == 0;

    // B operand: bqs[s][lane] = channel 2s+half of query q0+col (zero-padded).
    const int qcol = q0 + col;
    for (int s = wave; s < FH; s += FK_WAVES) {
        int c = 2 * s + half;
        bqs[s][lane] = (c < f && qcol < n) ? X[(size_t)c * n + qcol] : 0.f;
    }
    const float sq_q = qcol < n ? SQ[qcol] : 0.f;

    float rd[FK_QPW];
    int ri[FK_QPW], cnt[FK_QPW];
#pragma unroll
    for (int t = 0; t < FK_QPW; ++t) { rd[t] = INFINITY; ri[t] = 0x7fffffff; cnt[t] = 0; }

    for (int t0 = 0; t0 < n; t0 += FK_NC) {
        const int tn = min(FK_NC, n - t0);
        __syncthreads();                         // previous chunk's selection finished (and bqs ready)
        for (int c = threadIdx.x; c < FK_NC; c += FK_THREADS) sqc[c] = c < tn ? SQ[t0 + c] : 0.f;
        __syncthreads();
        // ---- Gram strip of this wave: candidates t0 + 128*wave + [0,128)
        const int strip = 128 * wave;
        if (strip < tn) {
            const int a4 = ((lane >> 3) & 3) * 32 + (lane & 7) * 4;     // lane's 4 candidates in the strip
            const int cand = t0 + strip + a4;
            const bool full = vec_ok && cand + 3 < n;
            const float *Ap = X + (size_t)half * n + (full ? cand : 0);
            f32x16 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            // Fast path (wave-uniform): float4 loads, no channel / candidate tail.  The loads run a
            // fixed PF steps ahead of the MFMAs through a register ring and the index is clamped
            // instead of branched, so the loop body is branch-free and the compiler keeps PF loads
            // in flight behind counted vmcnt waits.
            const bool fast = vec_ok && (t0 + strip + 128 <= n) && (2 * FH == f);
            if (fast) {
                constexpr int PF = 8;
                const float *Af = X + (size_t)half * n + cand;
                float4 ring[PF];
                float bring[PF];
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    ring[i] = *reinterpret_cast<const float4 *>(Af + (size_t)(2 * i) * n);
                    bring[i] = bqs[i][lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                for (int s0 = 0; s0 < FH; s0 += PF) {
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        const float4 v = ring[i];
                        const float bv = bring[i];
                        const int sn = min(s0 + PF + i, FH - 1);
                        ring[i] = *reinterpret_cast<const float4 *>(Af + (size_t)(2 * sn) * n);
                        bring[i] = bqs[sn][lane];
                        __builtin_amdgcn_sched_barrier(0);      // keep both loads PF steps ahead of their use
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, bv, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, bv, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, bv, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, bv, acc[3], 0, 0, 0);
                    }
                }
            } else
            for (int s = 0; s < FH; ++s) {
                const int c = 2 * s + half;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < f) {
                    if (full) {
                        v = *reinterpret_cast<const float4 *>(Ap + (size_t)(2 * s) * n);
                    } else {
                        const float *row = X + (size_t)c * n;
                        v.x = cand < n ? row[cand] : 0.f;
                        v.y = cand + 1 < n ? row[cand + 1] : 0.f;
                        v.z = cand + 2 < n ? row[cand + 2] : 0.f;
                        v.w = cand + 3 < n ? row[cand + 3] : 0.f;
                    }
                }
                const float bv = bqs[s][lane];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, bv, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, bv, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, bv, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, bv, acc[3], 0, 0, 0);
            }
            // D row i = (r&3) + 8*(r>>2) + 4*half  <->  (a = r>>2, b = (r&3) + 4*half): accumulator j,
            // register r holds candidate strip + 32a + 4b + j of query `col`.
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lc = strip + 32 * (r >> 2) + 16 * half + 4 * (r & 3);
                const float4 sc = *reinterpret_cast<const float4 *>(&sqc[lc]);
                float4 v;
                v.x = lc + 0 < tn ? __fmaf_rn(-2.0f, acc[0][r], sq_q) + sc.x : INFINITY;
                v.y = lc + 1 < tn ? __fmaf_rn(-2.0f, acc[1][r], sq_q) + sc.y : INFINITY;
                v.z = lc + 2 < tn ? __fmaf_rn(-2.0f, acc[2][r], sq_q) + sc.z : INFINITY;
                v.w = lc + 3 < tn ? __fmaf_rn(-2.0f, acc[3][r], sq_q) + sc.w : INFINITY;
                *reinterpret_cast<float4 *>(&dist[col][lc]) = v;
            }
        }
        __syncthreads();
        // ---- selection: wave w owns queries FK_QPW*w .. +FK_QPW-1
#pragma unroll
        for (int t = 0; t < FK_QPW; ++t) {
            const int ql = wave * FK_QPW + t;
            if (q0 + ql < n) {
                const float *row = dist[ql];
                wave_topk_append([&](int c) { return row[c]; }, tn, t0, queue[ql], cnt[t], K, rd[t], ri[t], lane);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < FK_QPW; ++t) {
        const int ql = wave * FK_QPW + t;
        const int qq = q0 + ql;
        if (qq < n) {
            knn_flush(queue[ql], cnt[t], K, rd[t], ri[t], lane);
            if (lane >= 1 && lane <= k)                       // rank 0 dropped (:458, :501)
                idx[((size_t)bs * n + qq) * k + lane - 1] = rd[t] < INFINITY ? ri[t] : 0;
        }
    }
}

// ---------------------------------------------------------------------------- producer / consumer form
// Regular shapes (n % 128 == 0, f == 2*FH): 12 waves per workgroup.  Waves 0-3 PRODUCE the Gram strips of
// chunk i on the matrix cores, waves 4-11 CONSUME (select from) chunk i-1; the two dist tiles alternate in LDS and
// one workgroup barrier per chunk hands them over.  Both operands are fed from global memory (L2-resident) through
// 8-deep register rings.
// Consumer waves.  Measured on gfx950 (tools/micro/mfma_ilp.hip, mfma_valu.hip): fp32 MFMAs execute on the SIMD's vector
// ALU -- a wave that issues them back to back locks every other wave of its SIMD out completely (no VALU, LDS or scalar
// instruction of theirs issues until the matrix stream stalls or ends; s_setprio does not change that), and inside one
// wave every vector instruction between two MFMAs adds its full 4+ cycles.  The selection therefore does NOT hide behind
// the Gram stream: per SIMD the chunk time is matrix time + selection time, the consumers running in the producer's
// operand-wait gaps and after its last MFMA.  What counts is the instruction count of the selection (wave_select.h: four
// queries interleaved per wave, scalar counters, no per-query branches, threshold + 64-bit-key ranking in the merge) and
// an MFMA loop without vector address arithmetic (buffer loads with scalar row offsets).  8 consumer waves (4 queries
// each) and 12 (3 each, 1024 threads, 128 VGPRs with spills) measure the same at N = 1024 and 8 are faster below.
#define FKP_CW 8
#define FKP_THREADS (256 + 64 * FKP_CW)
#define FKP_QPW (FK_QB / FKP_CW)  // queries per consumer wave (4: the row lists below)
static_assert(FK_QB / FKP_CW == 4, "the consumer's row lists are written for 4 queries per wave");
#define FKP_QCAP 64

#ifdef FK_TIMING
__device__ unsigned long long fk_dbg[8 * 64];
extern "C" int fk_dbg_copy(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fk_dbg), sizeof(fk_dbg)); }
#define FK_T(slot) do { if (blockIdx.x == 0 && lane == 0 && it < 64) fk_dbg[(slot) * 64 + it] = clock64(); } while (0)
#else
#define FK_T(slot) do { } while (0)
#endif

template <int FH>
__global__ __launch_bounds__(FKP_THREADS) void feat_knn_pc_kernel(
    int b, int n, int k, const float *__restrict__ x, const float *__restrict__ sq, int32_t *__restrict__ idx) {
    constexpr int NCP = FK_NC + 4;
    constexpr int f = 2 * FH;
    __shared__ float dist[2][FK_QB][NCP];
    __shared__ DI queue[FK_QB][FKP_QCAP + 32];       // + room for the running list during a ranked merge
    __shared__ DI win[FKP_CW][32];                     // per consumer wave

    // PERSISTENT workgroups, one per CU: workgroup ids go round-robin over the 8 XCDs; XCD x owns the samples x, x+8, ...
    // (every query tile of a sample streams that sample's whole feature matrix: one L2 fetches it once), and its
    // workgroups deal those samples' query tiles among themselves.  A workgroup walks its tiles as ONE stream of
    // 512-candidate chunks: while the consumer waves select from chunk i, the producer waves already build chunk i+1
    // -- of the same tile or the next one -- so the matrix cores idle only in the very first and last step.
    const int tiles = n / FK_QB, pid = blockIdx.x;
    const int xcd = pid & 7, slot = pid >> 3, wpx = gridDim.x >> 3;
    const int ns = (b - xcd + 7) / 8;                        // samples of this XCD
    const int T = ns > 0 ? ns * tiles : 0;                   // its query tiles
    const int mine = slot < T ? (T - slot + wpx - 1) / wpx : 0;
    if (mine == 0) return;
    const int lane = lane_id();
    const int wave = threadIdx.x / PDGN_WAVE;
    const bool producer = wave < 4;
    const int col = lane & 31, half = lane >> 5;
    const int K = k + 1;
    const int nchunks = (n + FK_NC - 1) / FK_NC;
    const int total = mine * nchunks;

    float rd[FKP_QPW];
    int ri[FKP_QPW], cnt[FKP_QPW];
#pragma unroll
    for (int t = 0; t < FKP_QPW; ++t) { rd[t] = INFINITY; ri[t] = 0x7fffffff; cnt[t] = 0; }

    for (int it = 0; it <= total; ++it) {
        if (producer) {
            const int j = slot + (it / nchunks) * wpx;            // tile index within the XCD
            const int bs = xcd + 8 * (j / tiles), q0 = (j % tiles) * FK_QB;
            const int t0 = (it % nchunks) * FK_NC;
            const int strip = 128 * wave;
            if (it < total && t0 + strip < n) {
                if (wave == 0) FK_T(0);
                const float *X = x + (size_t)bs * f * n;
                const float *SQ = sq + (size_t)bs * n;
                const float sq_q = SQ[q0 + col];
                const int a4 = ((lane >> 3) & 3) * 32 + (lane & 7) * 4;
                const int cand = t0 + strip + a4;
                // Operand rows through a raw buffer descriptor of the sample's feature matrix: the lane's byte offset is
                // fixed for the whole chunk and the channel-pair row is a SCALAR offset, so a step issues two loads and four
                // MFMAs and no vector address arithmetic (fp32 MFMA and VALU share the SIMD's issue slots).
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, f * n * 4, 0x00020000);
                const unsigned aoff = (unsigned)(half * n + cand) * 4u, boff = (unsigned)(half * n + q0 + col) * 4u;
                const unsigned rowb = 2u * (unsigned)n * 4u;                 // bytes between channel pairs
                f32x16 acc[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;
                constexpr int PF = FH >= 128 ? 8 : 4;
                f32x4 ring[PF];
                float bring[PF];
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    ring[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, aoff, (unsigned)i * rowb, 0));
                    bring[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, boff, (unsigned)i * rowb, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
#ifndef FK_ABLATE_MFMA
                for (int s0 = 0; s0 < FH; s0 += PF) {
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        const f32x4 v = ring[i];
                        const float bv = bring[i];
                        const unsigned so = (unsigned)min(s0 + PF + i, FH - 1) * rowb;
                        ring[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, aoff, so, 0));
                        bring[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, boff, so, 0));
                        __builtin_amdgcn_sched_barrier(0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, bv, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, bv, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, bv, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, bv, acc[3], 0, 0, 0);
                    }
                }
#endif
                if (wave == 0) FK_T(1);
                float (*D)[NCP] = dist[it & 1];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lc = strip + 32 * (r >> 2) + 16 * half + 4 * (r & 3);
                    const float4 sc = *reinterpret_cast<const float4 *>(SQ + t0 + lc);
                    float4 v;
                    v.x = __fmaf_rn(-2.0f, acc[0][r], sq_q) + sc.x;
                    v.y = __fmaf_rn(-2.0f, acc[1][r], sq_q) + sc.y;
                    v.z = __fmaf_rn(-2.0f, acc[2][r], sq_q) + sc.z;
                    v.w = __fmaf_rn(-2.0f, acc[3][r], sq_q) + sc.w;
                    *reinterpret_cast<float4 *>(&D[col][lc]) = v;
                }
                if (wave == 0) FK_T(2);
            }
        } else if (it >= 1) {
            if (wave == 4) FK_T(3);
            const int jt = (it - 1) / nchunks, ch = (it - 1) % nchunks;
            const int t0 = ch * FK_NC;
            const int tn = min(FK_NC, n - t0);
            const int cw = wave - 4, qb = cw * FKP_QPW;
            DI *const qs[FKP_QPW] = {queue[qb], queue[qb + 1], queue[qb + 2], queue[qb + 3]};
#ifndef FK_ABLATE_SELECT
            {
                const float *const rows[FKP_QPW] = {dist[(it - 1) & 1][qb], dist[(it - 1) & 1][qb + 1], dist[(it - 1) & 1][qb + 2],
                                                    dist[(it - 1) & 1][qb + 3]};
                wave_topk_append_multi<FKP_QPW>(rows, tn, t0, qs, cnt, FKP_QCAP, K, rd, ri, lane, [&](int t) {
                    knn_flush_ranked(qs[t], cnt[t], K, rd[t], ri[t], lane, win[cw]);
                    cnt[t] = 0;
                });
            }
#endif
            if (wave == 4) FK_T(4);
            if (ch == nchunks - 1) {                           // the tile's last chunk: merge, emit its rows, start afresh
                const int j = slot + jt * wpx;
                const int bs = xcd + 8 * (j / tiles), q0 = (j % tiles) * FK_QB;
                knn_flush_select_multi<FKP_QPW>(qs, cnt, K, rd, ri, lane, win[cw]);
                if (wave == 4) FK_T(7);
#pragma unroll
                for (int t = 0; t < FKP_QPW; ++t) {
                    if (lane >= 1 && lane <= k)
                        idx[((size_t)bs * n + q0 + qb + t) * k + lane - 1] = rd[t] < INFINITY ? ri[t] : 0;
                    rd[t] = INFINITY;
                    ri[t] = 0x7fffffff;
                    cnt[t] = 0;
                }
            }
            if (wave == 4) FK_T(5);
        }
        __syncthreads();
        if (wave == 0) FK_T(6);
    }
}

// Persistent grid of the producer / consumer kernel: one workgroup per CU (its 157 KB of LDS allow no more), a
// multiple of 8, no more per XCD than the busiest XCD has tiles.
static int fk_persistent_grid(int b, int tiles) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            cus = 256;
    }
    const int per_xcd_tiles = ((b + 7) / 8) * tiles;
    int wpx = cus / 8;
    if (wpx > per_xcd_tiles) wpx = per_xcd_tiles;
    return 8 * wpx;
}

template <int FH>
static int launch_fk(int b, int f, int n, int k, const float *x, const float *sq, int32_t *idx,
                     hipStream_t s) {
    dim3 grid(cdiv(n, FK_QB), b);
    if (f == 2 * FH && n % 128 == 0)
        hipLaunchKernelGGL((feat_knn_pc_kernel<FH>), dim3((unsigned)fk_persistent_grid(b, n / FK_QB)), dim3(FKP_THREADS), 0, s, b,
                           n, k, x, sq, idx);
    else
        hipLaunchKernelGGL((feat_knn_kernel<FH>), grid, dim3(FK_THREADS), 0, s, f, n, k, x, sq, idx);
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
