// gemm_nt.hip -- the dense contraction of the point-major layers on the fp32 matrix cores.
//
//   C[m, n] = sum_k A[m, k] * W[n, k]  (+ bias[n]) (+ addend[m, n])        A (M x K), W (N x K) row-major
//
// i.e. rows x C_in @ (C_out x C_in)^T -- the shape of every conv / linear of the deconvolution
// stack once activations are point-major (the per-point GEMM Y = X Wcat^T, conv_all, the
// inte*w contraction of conv2, MLP heads, discriminators), and of their input gradients
// dX = dY W (call it with W^T).  Optional epilogue: per-column partial sums / sums of squares of
// the block's rows, so the BatchNorm that follows needs no statistics pass over C.
//
// Tiling (wave64, v_mfma_f32_32x32x2_f32 = exact fp32 fma chains):
//   * workgroup 256 threads = 2x2 waves, tile 128 x 128, K chunk 32; each wave 64 x 64 = 2x2
//     accumulators of 32x32 (64 accumulator registers);
//   * both operands are K-contiguous, so tiles are copied global -> LDS untransposed with float4
//     loads/stores (row pitch 36 floats: conflict-free ds_read_b128);
//   * a lane feeds FOUR consecutive MFMA k-steps from ONE ds_read_b128 per operand: lane (i, h)
//     holds A[i][8q+4h .. 8q+4h+3]; step u multiplies element u of both operands -- the MFMA sums over
//     k, so any assignment of k values to (step, half) is valid as long as A and B agree.
//     16 MFMAs per 4 LDS reads;
//   * the next K chunk is prefetched into registers while the current one is multiplied.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NT_THREADS 256
#define NT_BM 128
#define NT_BN 128
#define NT_BK 32
#define NT_LD 36

__global__ __launch_bounds__(NT_THREADS) void gemm_nt_kernel(
    long long M, int N, int K, const float *__restrict__ A, const float *__restrict__ W,
    const float *__restrict__ bias, const float *__restrict__ addend, float *__restrict__ C,
    float *__restrict__ stat_part) {
    __shared__ float As[NT_BM][NT_LD];
    __shared__ float Bs[NT_BN][NT_LD];
    // blockIdx.x walks the N tiles fastest: consecutive workgroups share the same A rows (L2 reuse)
    const int n0 = blockIdx.x * NT_BN;
    const long long m0 = (long long)blockIdx.y * NT_BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;                 // wave's 64x64 quadrant
    const int li = lane & 31, half = lane >> 5;

    // staging: thread loads float4 (row r_ld + 32*i, k-offset k4), i = 0..3, for A and W
    const int r_ld = tid >> 3, k4 = (tid & 7) * 4;
    float4 pa[4], pb[4];
    auto prefetch = [&](int kk) {
        const bool kok = kk + k4 < K;                         // K % 4 == 0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long m = m0 + r_ld + 32 * i;
            const int n = n0 + r_ld + 32 * i;
            pa[i] = (kok && m < M) ? *reinterpret_cast<const float4 *>(A + m * K + kk + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
            pb[i] = (kok && n < N) ? *reinterpret_cast<const float4 *>(W + (size_t)n * K + kk + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    prefetch(0);
    for (int kk = 0; kk < K; kk += NT_BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4 *>(&As[r_ld + 32 * i][k4]) = pa[i];
            *reinterpret_cast<float4 *>(&Bs[r_ld + 32 * i][k4]) = pb[i];
        }
        __syncthreads();
        if (kk + NT_BK < K) prefetch(kk + NT_BK);
#pragma unroll
        for (int q = 0; q < NT_BK / 8; ++q) {
            float4 av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[t] = *reinterpret_cast<const float4 *>(&As[wr * 64 + t * 32 + li][8 * q + 4 * half]);
                bv[t] = *reinterpret_cast<const float4 *>(&Bs[wc * 64 + t * 32 + li][8 * q + 4 * half]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].x, bv[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].y, bv[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].z, bv[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].w, bv[b].w, acc[a][b], 0, 0, 0);
                }
        }
    }
    // epilogue: D[row = (r&3) + 8*(r>>2) + 4*half][col = li]
    float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wc * 64 + b * 32 + li;
        const bool nok = n < N;
        const float bz = (bias && nok) ? bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (nok && m < M) {
                    float v = acc[a][b][r] + bz;
                    if (addend) v += addend[m * N + n];
                    C[m * N + n] = v;
                    csum[b] += v;
                    csq[b] = __fmaf_rn(v, v, csq[b]);
                }
            }
        }
    }
    if (stat_part) {
        // combine the two lane halves, then the two row-halves of the workgroup through LDS
        __syncthreads();
        float *red = &As[0][0];                               // reuse: [wr][2][128]
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float s = csum[b] + __shfl_xor(csum[b], 32, 64);
            float q = csq[b] + __shfl_xor(csq[b], 32, 64);
            if (half == 0) {
                const int cl = wc * 64 + b * 32 + li;
                red[(wr * 2 + 0) * 128 + cl] = s;
                red[(wr * 2 + 1) * 128 + cl] = q;
            }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < N) {
            float *P = stat_part + (size_t)blockIdx.y * 2 * N;
            P[n0 + tid] = red[0 * 128 + tid] + red[2 * 128 + tid];
            P[N + n0 + tid] = red[1 * 128 + tid] + red[3 * 128 + tid];
        }
    }
}

// C (m x n) = A (m x k) W (n x k)^T (+ bias) (+ addend); k % 4 == 0.  stat_part (may be NULL):
// ceil(m/128) rows of [2n] floats receiving the per-column sum / sum of squares of each 128-row block
// (the layout pdgn_bn_finalize consumes).
extern "C" int pdgn_gemm_nt(long long m, int n, int k, const float *A, const float *W, const float *bias,
                            const float *addend, float *C, float *stat_part, pdgn_stream_t stream) {
    if (m < 1 || n < 1 || k < 4 || k % 4) return PDGN_ERR_INVALID;
    const long long gy = (m + NT_BM - 1) / NT_BM;
    if (gy > 65535) return PDGN_ERR_INVALID;
    dim3 grid(cdiv(n, NT_BN), (unsigned)gy);
    hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(NT_THREADS), 0, (hipStream_t)stream, m, n, k, A, W, bias, addend, C,
                       stat_part);
    return pdgn_launch_status();
}
