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
// Tiling: workgroup = 4 waves, output tile 128 (n) x 128 (k), reduction chunk 32 rows.
//   * both operands are copied global -> LDS untransposed ([row m][channel], channel contiguous):
//     coalesced 512-byte row segments in, float4 LDS writes;
//   * MFMA v_mfma_f32_32x32x2_f32 with the reduction rows on the MFMA k index: lane (h,a,b) reads ONE
//     float4 of the dY tile, As[2s+h][32a+4b .. +3], whose element j is the A operand of accumulator
//     j (output rows 32a+4b+j), and one float of the X tile, Bs[2s+h][32*wave + lane&31]: four MFMAs
//     per (ds_read_b128 + ds_read_b32), 64 accumulator registers per lane;
//   * the next chunk is prefetched into registers while the current one is multiplied.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TN_THREADS 256
#define TN_BN 128          // output rows per workgroup  (dY columns)
#define TN_BK 128          // output cols per workgroup  (X columns)
#define TN_BM 32           // reduction rows per chunk
#define TN_LD (128 + 4)    // LDS row pitch (floats)

__global__ __launch_bounds__(TN_THREADS) void gemm_tn_kernel(
    long long M, int N, int K, long long rows_per_split, int use_atomic, const float *__restrict__ dY,
    const float *__restrict__ X, float *__restrict__ dW) {
    __shared__ float As[TN_BM][TN_LD];
    __shared__ float Bs[TN_BM][TN_LD];
    const int n0 = blockIdx.x * TN_BN, k0 = blockIdx.y * TN_BK;
    const long long m_begin = (long long)blockIdx.z * rows_per_split;
    const long long m_end = min(M, m_begin + rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int a4 = ((lane >> 3) & 3) * 32 + (lane & 7) * 4;

    // staging map: thread loads float4 (row r_ld + 8*i, cols c4..c4+3), i = 0..3
    const int r_ld = tid >> 5, c4 = (tid & 31) * 4;
    const bool an_ok = n0 + c4 < N, bk_ok = k0 + c4 < K;      // N, K are multiples of 4
    float4 pa[4], pb[4];
    auto prefetch = [&](long long m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long m = m0 + r_ld + 8 * i;
            const bool ok = m < m_end;
            pa[i] = (ok && an_ok) ? *reinterpret_cast<const float4 *>(dY + m * N + n0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            pb[i] = (ok && bk_ok) ? *reinterpret_cast<const float4 *>(X + m * K + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    prefetch(m_begin);
    for (long long m0 = m_begin; m0 < m_end; m0 += TN_BM) {
        __syncthreads();                                         // previous chunk consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4 *>(&As[r_ld + 8 * i][c4]) = pa[i];
            *reinterpret_cast<float4 *>(&Bs[r_ld + 8 * i][c4]) = pb[i];
        }
        __syncthreads();
        if (m0 + TN_BM < m_end) prefetch(m0 + TN_BM);            // overlaps the MFMAs below
#pragma unroll
        for (int s = 0; s < TN_BM / 2; ++s) {
            const float4 av = *reinterpret_cast<const float4 *>(&As[2 * s + half][a4]);
            const float bv = Bs[2 * s + half][wave * 32 + col];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv, acc[3], 0, 0, 0);
        }
    }
    // D row i = (r&3) + 8*(r>>2) + 4*half <-> (a = r>>2, b = (r&3) + 4*half): accumulator j, register r
    // holds output row n0 + 32a + 4b + j, column k0 + 32*wave + col.
    const int kc = k0 + wave * 32 + col;
    if (kc < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + 32 * (r >> 2) + 4 * ((r & 3) + 4 * half) + j;
                if (n < N) {
                    float *dst = dW + (size_t)n * K + kc;
                    if (use_atomic) atomicAdd(dst, acc[j][r]);
                    else *dst = acc[j][r];
                }
            }
        }
    }
}

// dW (N x K) = dY^T X.  dW must be ZERO-FILLED by the caller when the launch splits M (it always
// may: zero-fill unconditionally).  N % 4 == 0, K % 4 == 0.
extern "C" int pdgn_gemm_tn(long long m, int n, int k, const float *dY, const float *X, float *dW,
                            pdgn_stream_t stream) {
    if (m < 1 || n < 4 || k < 4 || n % 4 || k % 4) return PDGN_ERR_INVALID;
    const int gx = cdiv(n, TN_BN), gy = cdiv(k, TN_BK);
    long long splits = 1024 / ((long long)gx * gy);              // ~4 workgroups per CU
    const long long max_splits = (m + 8 * TN_BM - 1) / (8 * TN_BM);
    splits = splits < 1 ? 1 : (splits > max_splits ? max_splits : splits);
    splits = splits > 65535 ? 65535 : splits;
    long long rows = (m + splits - 1) / splits;
    rows = (rows + TN_BM - 1) / TN_BM * TN_BM;
    splits = (m + rows - 1) / rows;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(gx, gy, (unsigned)splits), dim3(TN_THREADS), 0, (hipStream_t)stream, m, n, k,
                       rows, splits > 1 ? 1 : 0, dY, X, dW);
    return pdgn_launch_status();
}
