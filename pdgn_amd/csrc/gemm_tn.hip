// gemm_tn.hip -- weight-gradient GEMM of the point-major dense layers on the fp32 matrix cores.
//
//   dW[n, k] (+)= sum_m dY[m, n] * X[m, k]          dY (M x N), X (M x K) row-major, M >> N, K
//
// Every dense layer of the deconvolution stack / discriminators is rows x C_in @ (C_out x C_in)^T
// with 10^4..10^5.5 rows, so its weight gradient is a "TN" product whose reduction runs over the
// row axis and whose output is small.  Library GEMMs do not split that reduction (measured 5-55
// TFLOP/s on these shapes); here the M axis is split over workgroups and the partial tiles are
// combined with fp32 atomics into the (pre-zeroed) output.
//
// Tiling: workgroup = 4 waves arranged WN (along n) x WK (along k); a wave owns 128 output rows x
// 32*KG output columns = 4*KG accumulators of 32x32; workgroup tile (128*WN) x (32*WK*KG);
// reduction chunk 32 rows.  Variants (picked by K): 512x32, 256x64, 128x128, 128x256.
//   * both operands are copied global -> LDS untransposed ([row m][channel], channel contiguous):
//     coalesced row segments in, float4 LDS writes;
//   * MFMA v_mfma_f32_32x32x2_f32 with the reduction rows on the MFMA k index: lane (h,a,b) reads ONE
//     float4 of the dY tile, As[2s+h][32a+4b .. +3], whose element j is the A operand of accumulator
//     j (output rows 32a+4b+j), and KG floats of the X tile: 4*KG MFMAs per (ds_read_b128 + KG ds_read_b32);
//   * the next chunk is prefetched into registers while the current one is multiplied;
//   * the 1-D grid is decoded XCD-aware: workgroup ids are dealt round-robin to the 8 XCDs, so id ->
//     (xcd, slot) -> a contiguous range of tiles per XCD, n-tile fastest.  Tiles that stream the same
//     X column strip (same k-tile, same split) or the same dY strip therefore sit behind ONE L2 and
//     move through the rows in lockstep: each strip leaves HBM once instead of once per XCD.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TN_THREADS 256
#define TN_BM 32           // reduction rows per chunk

template <int WN, int WK, int KG>
__global__ __launch_bounds__(TN_THREADS) void gemm_tn_kernel(
    long long M, int N, int K, long long rows_per_split, int use_atomic, int gx, int gy,
    const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ dW) {
    constexpr int TBN = 128 * WN, TBK = 32 * WK * KG;
    constexpr int LDA = TBN + 4, LDB = TBK + 4;
    constexpr int NA = TBN / 32, NB = TBK / 32;               // float4 per thread per chunk (>= 1)
    static_assert(WN * WK == 4, "four waves");
    __shared__ float As[TN_BM][LDA];
    __shared__ float Bs[TN_BM][LDB];

    // XCD-aware decode of the 1-D grid
    const int total = gridDim.x, pid = blockIdx.x;
    const int q = total >> 3, r8 = total & 7, xcd = pid & 7, slot = pid >> 3;
    const int v = xcd * q + min(xcd, r8) + slot;
    const int bx = v % gx, by = (v / gx) % gy, bz = v / (gx * gy);

    const int n0 = bx * TBN, k0 = by * TBK;
    const long long m_begin = (long long)bz * rows_per_split;
    const long long m_end = min(M, m_begin + rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave % WK, wn = wave / WK;
    const int half = lane >> 5, col = lane & 31;
    const int a4 = wn * 128 + ((lane >> 3) & 3) * 32 + (lane & 7) * 4;

    // staging map: float4 index tid + 256*i of the [32][TBN/4] (resp. [32][TBK/4]) tile.  Columns past N / K
    // are clamped to the last valid float4: they only feed accumulator rows / columns that are never
    // stored.  Rows need a guard in the last (partial) chunk of the last split only, so the steady-state
    // prefetch is branch-free: with two workgroups per SIMD running the same barrier-paced program in
    // lockstep, every instruction between the barrier and the first MFMA is idle matrix-core time.
    const int ar = tid / (TBN / 4), ac = min(n0 + (tid % (TBN / 4)) * 4, N - 4);
    const int br = tid / (TBK / 4), bc = min(k0 + (tid % (TBK / 4)) * 4, K - 4);
    const int lac = (tid % (TBN / 4)) * 4, lbc = (tid % (TBK / 4)) * 4;
    constexpr int AR_STEP = 256 / (TBN / 4), BR_STEP = 256 / (TBK / 4);
    const float *pA = dY + (m_begin + ar) * N + ac;
    const float *pB = X + (m_begin + br) * K + bc;
    float4 pa[NA], pb[NB];
    auto prefetch_full = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) pa[i] = *reinterpret_cast<const float4 *>(pA + (long long)(AR_STEP * i) * N);
#pragma unroll
        for (int i = 0; i < NB; ++i) pb[i] = *reinterpret_cast<const float4 *>(pB + (long long)(BR_STEP * i) * K);
    };
    auto prefetch_tail = [&](long long m0) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            pa[i] = (m0 + ar + AR_STEP * i < m_end) ? *reinterpret_cast<const float4 *>(pA + (long long)(AR_STEP * i) * N)
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            pb[i] = (m0 + br + BR_STEP * i < m_end) ? *reinterpret_cast<const float4 *>(pB + (long long)(BR_STEP * i) * K)
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    f32x16 acc[KG][4];
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][j][r] = 0.f;

    const long long span = m_end - m_begin;
    const int nfull = (int)(span / TN_BM), nchunk = (int)((span + TN_BM - 1) / TN_BM);
    if (nfull > 0) prefetch_full(); else prefetch_tail(m_begin);
    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();                                         // previous chunk consumed
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<float4 *>(&As[ar + AR_STEP * i][lac]) = pa[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4 *>(&Bs[br + BR_STEP * i][lbc]) = pb[i];
        __syncthreads();
        pA += (long long)TN_BM * N;
        pB += (long long)TN_BM * K;
        if (c + 1 < nfull) prefetch_full();                      // overlaps the MFMAs below
        else if (c + 1 < nchunk) prefetch_tail(m_begin + (long long)(c + 1) * TN_BM);
        // software-pipelined operand reads: step s+1's LDS reads are issued BEFORE step s's MFMAs
        float4 av = *reinterpret_cast<const float4 *>(&As[half][a4]);
        float bv[KG];
#pragma unroll
        for (int g = 0; g < KG; ++g) bv[g] = Bs[half][(wk * KG + g) * 32 + col];
#pragma unroll
        for (int s = 0; s < TN_BM / 2; ++s) {
            float4 an = av;
            float bn[KG];
#pragma unroll
            for (int g = 0; g < KG; ++g) bn[g] = bv[g];
            if (s + 1 < TN_BM / 2) {
                an = *reinterpret_cast<const float4 *>(&As[2 * s + 2 + half][a4]);
#pragma unroll
                for (int g = 0; g < KG; ++g) bn[g] = Bs[2 * s + 2 + half][(wk * KG + g) * 32 + col];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[g], acc[g][0], 0, 0, 0);
                acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[g], acc[g][1], 0, 0, 0);
                acc[g][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[g], acc[g][2], 0, 0, 0);
                acc[g][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[g], acc[g][3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            av = an;
#pragma unroll
            for (int g = 0; g < KG; ++g) bv[g] = bn[g];
        }
    }
    // D row i = (r&3) + 8*(r>>2) + 4*half <-> (a = r>>2, b = (r&3) + 4*half): accumulator j, register r
    // holds output row n0 + 128*wn + 32a + 4b + j, column k0 + 32*(wk*KG+g) + col.
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        const int kc = k0 + (wk * KG + g) * 32 + col;
        if (kc < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 128 + 32 * (r >> 2) + 4 * ((r & 3) + 4 * half) + j;
                    if (n < N) {
                        float *dst = dW + (size_t)n * K + kc;
                        if (use_atomic) atomicAdd(dst, acc[g][j][r]);
                        else *dst = acc[g][j][r];
                    }
                }
            }
        }
    }
}

// Tiny outputs (N*K <= 1024: the xyz input layers, K = 3 padded to 4, and the 16-channel bilateral
// branch): an MFMA tile would be >= 87 % padding, and the product is a pure stream over dY.  One thread per
// output element (4 per thread at 1024), rows staged through LDS 64 at a time, one atomic per output
// and workgroup.
#define TS_ROWS 64
__global__ __launch_bounds__(256) void gemm_tn_small_kernel(long long M, int N, int K, long long rows_per_wg,
                                                            const float *__restrict__ dY, const float *__restrict__ X,
                                                            float *__restrict__ dW) {
    extern __shared__ float sm[];
    float *Ys = sm, *Xs = sm + TS_ROWS * N;                     // [64][N], [64][K]
    const int tid = threadIdx.x, NK = N * K;
    const long long m_begin = (long long)blockIdx.x * rows_per_wg, m_end = min(M, m_begin + rows_per_wg);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int on[4], ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = min(tid + 256 * j, NK - 1);
        on[j] = o / K;
        ok[j] = o % K;
    }
    for (long long m0 = m_begin; m0 < m_end; m0 += TS_ROWS) {
        const int rows = (int)min((long long)TS_ROWS, m_end - m0);
        __syncthreads();
        for (int i = tid * 4; i < rows * N; i += 1024)
            *reinterpret_cast<float4 *>(Ys + i) = *reinterpret_cast<const float4 *>(dY + m0 * N + i);
        for (int i = tid * 4; i < rows * K; i += 1024)
            *reinterpret_cast<float4 *>(Xs + i) = *reinterpret_cast<const float4 *>(X + m0 * K + i);
        __syncthreads();
        for (int r = 0; r < rows; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __fmaf_rn(Ys[r * N + on[j]], Xs[r * K + ok[j]], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (tid + 256 * j < NK) atomicAdd(dW + tid + 256 * j, acc[j]);
}

// Workgroups that fit on the chip at once (occupancy x CUs), per variant, queried once.
template <int WN, int WK, int KG>
static int tn_resident() {
    static int cached = 0;
    if (!cached) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_tn_kernel<WN, WK, KG>, TN_THREADS, 0) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1)
            return 512;
        cached = (per_cu > 2 ? 2 : per_cu) * cus;                // a third co-resident workgroup measured slower
    }
    return cached;
}

template <int WN, int WK, int KG>
static int tn_launch(long long m, int n, int k, const float *dY, const float *X, float *dW, hipStream_t stream) {
    constexpr int TBN = 128 * WN, TBK = 32 * WK * KG;
    const long long g = (long long)cdiv(n, TBN) * cdiv(k, TBK);
    const int gx = cdiv(n, TBN), gy = cdiv(k, TBK);
    // Split M so that the launch fills whole "rounds" of resident workgroups: a 1.3-round launch costs two rounds.
    const long long cap = tn_resident<WN, WK, KG>();
    const long long max_splits = (m + 8 * TN_BM - 1) / (8 * TN_BM);
    long long hi = 8 * cap / g + 1;
    hi = hi > max_splits ? max_splits : hi;
    long long splits = 1;
    double best = -1.0;
    for (long long sp = 1; sp <= hi; ++sp) {
        const long long tot = g * sp, rounds = (tot + cap - 1) / cap;
        const double eff = (double)tot / (double)(rounds * cap);
        if (eff > best + 0.02) { best = eff; splits = sp; }       // prefer fewer splits (fewer atomics) unless clearly better
        if (best >= 0.92) break;
    }
    long long rows = (m + splits - 1) / splits;
    rows = (rows + TN_BM - 1) / TN_BM * TN_BM;
    splits = (m + rows - 1) / rows;
    const long long total = g * splits;
    if (total > 0x7fffffffLL) return PDGN_ERR_INVALID;
    hipLaunchKernelGGL((gemm_tn_kernel<WN, WK, KG>), dim3((unsigned)total), dim3(TN_THREADS), 0, stream, m, n, k, rows,
                       splits > 1 ? 1 : 0, gx, gy, dY, X, dW);
    return pdgn_launch_status();
}

// dW (N x K) = dY^T X.  dW must be ZERO-FILLED by the caller when the launch splits M (it always
// may: zero-fill unconditionally).  N % 4 == 0, K % 4 == 0.
extern "C" int pdgn_gemm_tn(long long m, int n, int k, const float *dY, const float *X, float *dW,
                            pdgn_stream_t stream) {
    if (m < 1 || n < 4 || k < 4 || n % 4 || k % 4) return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if ((long long)n * k <= 1024) {
        long long wgs = (m + 4 * TS_ROWS - 1) / (4 * TS_ROWS);
        wgs = wgs > 2048 ? 2048 : wgs;
        long long rows = (m + wgs - 1) / wgs;
        rows = (rows + TS_ROWS - 1) / TS_ROWS * TS_ROWS;
        wgs = (m + rows - 1) / rows;
        hipLaunchKernelGGL(gemm_tn_small_kernel, dim3((unsigned)wgs), dim3(256), (size_t)TS_ROWS * (n + k) * sizeof(float), s, m,
                           n, k, rows, dY, X, dW);
        return pdgn_launch_status();
    }
    if (k <= 32) return tn_launch<4, 1, 1>(m, n, k, dY, X, dW, s);      // 512 x 32
    if (k <= 64) return tn_launch<2, 2, 1>(m, n, k, dY, X, dW, s);      // 256 x 64
    if (k <= 128) return tn_launch<1, 4, 1>(m, n, k, dY, X, dW, s);     // 128 x 128
    return tn_launch<1, 4, 2>(m, n, k, dY, X, dW, s);                    // 128 x 256
}
