// skinny.hip -- the products of the per-sample layers: one operand has R <= 64 rows (the batch: 35 per GPU).
//
// The generator's per-sample vectors (models/PDGNet_v2.py:704-707 xs / g, :825-828 fc1) meet the large weights in three
// forms, forward and backward:
//   nt:  C (R x N)   = A (R x K) B (N x K)^T (+ bias)     the constant-channel contribution Yc = const WcatC^T (DESIGN.md
//                                                          section 2), the heads' per-sample row g W0[:, :512]^T + b0 (:835-862)
//   nn:  C (R x N)  += A (R x K) B (K x N)                 their input gradients (dconst = dYc WcatC, dg = drb W0[:, :512]) and
//                                                          the dx = dpre W of the fused small layers (small_mlp.hip)
//   tn:  C (N x K)   = A (R x N)^T B (R x K)               their weight gradients (dWcatC = dYc^T const, dW0[:, :512] = drb^T g)
// Every one of them streams ONE large matrix (0.1 .. 6.6 MB) once against a 35-row operand: bandwidth, not arithmetic.  They ran
// on the BLAS library's skinny solutions through torch (53 launches per iteration, 15 .. 34 us each for 1 .. 7 us of traffic:
// tiles of 256 x 48 for a 35 x 256 result) -- the last matrix code of the iteration that was not written here.  These kernels
// keep the fp32 matrix instruction (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation) with the 35 rows padded to
// 48 / 64 in registers only: a wave owns 16 columns of the large operand, reads 16 B per lane along its contiguous axis, and
// the k order inside a 16-deep step is permuted identically on both operands so that each lane's float4 feeds four MFMA steps.
#include "common.h"

typedef float sk_f32x4 __attribute__((ext_vector_type(4)));

#define SK_THREADS 256
#define SK_MAXB 4                      // row blocks of 16: R <= 64

__device__ __forceinline__ sk_f32x4 sk_mfma(float a, float b, sk_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------- nt: C[r][n] = bias[n] + sum_k A[r][k] B[n][k]
// One wave = 16 columns n; lane (i = l % 16, q = l / 16) loads float4 B[n0 + i][k0 + 4q ..] and float4 A[16 blk + i][k0 + 4q ..]
// per 16-deep step; MFMA step s uses element s of both (k = k0 + 4q + s on both sides).  SPLIT: the four waves of a workgroup
// share 16 columns and split K (few columns: more workgroups, shorter loops), partial tiles joined through LDS; otherwise each
// wave has its own 16 columns (64 per workgroup).
template <bool SPLIT>
__global__ __launch_bounds__(SK_THREADS) void skinny_nt_kernel(int R, int N, int K, const float *__restrict__ A, int lda,
                                                               const float *__restrict__ B, int ldb,
                                                               const float *__restrict__ bias, float *__restrict__ C, int ldc, int act) {
    __shared__ float red[SPLIT ? 3 * SK_MAXB * 256 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int nb = (R + 15) >> 4;
    const int n0 = SPLIT ? blockIdx.x * 16 : (blockIdx.x * 4 + wave) * 16;
    const int kchunks = (K + 15) >> 4;
    int c0 = 0, c1 = kchunks;
    if (SPLIT) {
        const int per = (kchunks + 3) >> 2;
        c0 = wave * per;
        c1 = min(kchunks, c0 + per);
    }
    sk_f32x4 acc[SK_MAXB];
#pragma unroll
    for (int b = 0; b < SK_MAXB; ++b) acc[b] = (sk_f32x4){0.f, 0.f, 0.f, 0.f};
    const bool nok = n0 + i < N;
    const float *brow = B + (size_t)(nok ? n0 + i : 0) * ldb;
    for (int c = c0; c < c1; ++c) {
        const int k = c * 16 + 4 * q;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nok && k < K) bv = *reinterpret_cast<const float4 *>(brow + k);            // K % 4 == 0: a quad is all in or all out
        float4 av[SK_MAXB];
#pragma unroll
        for (int b = 0; b < SK_MAXB; ++b) {
            const int r = 16 * b + i;
            av[b] = (b < nb && r < R && k < K) ? *reinterpret_cast<const float4 *>(A + (size_t)r * lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int b = 0; b < SK_MAXB; ++b) {
            if (b < nb) {
                acc[b] = sk_mfma(av[b].x, bv.x, acc[b]);
                acc[b] = sk_mfma(av[b].y, bv.y, acc[b]);
                acc[b] = sk_mfma(av[b].z, bv.z, acc[b]);
                acc[b] = sk_mfma(av[b].w, bv.w, acc[b]);
            }
        }
    }
    if (SPLIT) {
        if (wave > 0) {
#pragma unroll
            for (int b = 0; b < SK_MAXB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[((wave - 1) * SK_MAXB + b) * 256 + j * 64 + lane] = acc[b][j];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int b = 0; b < SK_MAXB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[b][j] += red[(w * SK_MAXB + b) * 256 + j * 64 + lane];
    }
    // D[4q + j][i] of block b = C[16 b + 4 q + j][n0 + i]
    if (!nok) return;
    const float bz = bias ? bias[n0 + i] : 0.f;
#pragma unroll
    for (int b = 0; b < SK_MAXB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 16 * b + 4 * q + j;
            if (b < nb && r < R) {
                const float z = acc[b][j] + bz;
                C[(size_t)r * ldc + n0 + i] = act == 2 ? (z > 0.f ? z : 0.01f * z) : (act == 1 ? fmaxf(z, 0.f) : z);
            }
        }
}

// ---------------------------------------------------------------- nn: C[r][n] += sum_k A[r][k] B[k][n]     (C zero-filled by the caller)
// A workgroup = 64 columns n0 .. n0 + 63 and one slice of K; its four waves take every fourth 16-deep step of the slice.  Lane
// (u = l % 16, q = l / 16) loads float4 B[k0 + 4q + s][n0 + 4u ..] (s = 0 .. 3) and float4 A[16 blk + u][k0 + 4q ..]; column block e
// of the wave's output holds the columns n0 + 4u + e.  Partial sums leave by fp32 atomics (several slices per column block).
// MASK: A is taken as A[r][k] * act'(P[r][k]) (P = the pre-activation a small layer saved, act 1 = ReLU, 2 = LeakyReLU(0.01)): the
// input gradient of a frozen Linear + activation in one launch (small_mlp.hip's backward writes that product for the weight
// gradient; with frozen weights nobody else reads it).
template <bool MASK>
__global__ __launch_bounds__(SK_THREADS) void skinny_nn_kernel(int R, int N, int K, int chunks_per_slice, const float *__restrict__ A,
                                                               int lda, const float *__restrict__ B, int ldb, float *__restrict__ C,
                                                               int ldc, const float *__restrict__ P, int ldp, int act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = lane & 15, q = lane >> 4;
    const int nb = (R + 15) >> 4;
    const int n0 = blockIdx.x * 64;
    const int kchunks = (K + 15) >> 4;
    const int c0 = blockIdx.y * chunks_per_slice, c1 = min(kchunks, c0 + chunks_per_slice);
    sk_f32x4 acc[SK_MAXB][4];
#pragma unroll
    for (int b = 0; b < SK_MAXB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[b][e] = (sk_f32x4){0.f, 0.f, 0.f, 0.f};
    const bool nok = n0 + 4 * u < N;                                                  // N % 4 == 0
    for (int c = c0 + wave; c < c1; c += 4) {
        const int k = c * 16 + 4 * q;
        float bv[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nok && k + s < K) v = *reinterpret_cast<const float4 *>(B + (size_t)(k + s) * ldb + n0 + 4 * u);
            bv[s][0] = v.x; bv[s][1] = v.y; bv[s][2] = v.z; bv[s][3] = v.w;
        }
#pragma unroll
        for (int b = 0; b < SK_MAXB; ++b) {
            if (b < nb) {
                const int r = 16 * b + u;
                float av[4] = {0.f, 0.f, 0.f, 0.f};
                if (r < R) {
                    if (k + 3 < K) {
                        const float4 v = *reinterpret_cast<const float4 *>(A + (size_t)r * lda + k);
                        av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) av[s] = k + s < K ? A[(size_t)r * lda + k + s] : 0.f;
                    }
                    if (MASK) {
                        const float neg = act == 2 ? 0.01f : 0.f;
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            if (k + s < K && !(P[(size_t)r * ldp + k + s] > 0.f)) av[s] *= neg;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[b][e] = sk_mfma(av[s], bv[s][e], acc[b][e]);
            }
        }
    }
    // the four waves' partial tiles are joined in LDS; one set of stores (a single K slice) or atomics per workgroup
    __shared__ float red[3 * SK_MAXB * 4 * 256];
    if (wave > 0) {
#pragma unroll
        for (int b = 0; b < SK_MAXB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[(((wave - 1) * SK_MAXB + b) * 4 + e) * 256 + j * 64 + lane] = acc[b][e][j];
    }
    __syncthreads();
    if (wave > 0 || !nok) return;
    const bool single = gridDim.y == 1;
#pragma unroll
    for (int b = 0; b < SK_MAXB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 16 * b + 4 * q + j;
                if (b < nb && r < R) {
                    float v = acc[b][e][j];
#pragma unroll
                    for (int w = 0; w < 3; ++w) v += red[((w * SK_MAXB + b) * 4 + e) * 256 + j * 64 + lane];
                    float *dst = C + (size_t)r * ldc + n0 + 4 * u + e;
                    if (single) *dst = v;
                    else atomicAdd(dst, v);
                }
            }
}

// ---------------------------------------------------------------- tn: C[n][k] = sum_r A[r][n] B[r][k]
// A wave = 16 rows n of the output; the reduction (R <= 64, padded to a multiple of 4 with zeros) is 4 .. 16 MFMA steps whose A
// fragments (A[4 s + q][n0 + i]) stay in registers while the wave walks the 16-column blocks of k.
__global__ __launch_bounds__(SK_THREADS) void skinny_tn_kernel(int R, int N, int K, const float *__restrict__ A, int lda,
                                                               const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= N) return;
    const int steps = (R + 3) >> 2;
    float af[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int r = 4 * s + q;
        af[s] = (s < steps && r < R && n0 + i < N) ? A[(size_t)r * lda + n0 + i] : 0.f;
    }
    const int kblocks = (K + 15) >> 4;
    for (int kb = blockIdx.y; kb < kblocks; kb += gridDim.y) {
        const int k = kb * 16 + i;
        sk_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s < steps) {
                const int r = 4 * s + q;
                const float b = (r < R && k < K) ? B[(size_t)r * ldb + k] : 0.f;
                acc = sk_mfma(af[s], b, acc);
            }
        }
        if (k < K) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + 4 * q + j;
                if (n < N) C[(size_t)n * ldc + k] = acc[j];
            }
        }
    }
}

static bool sk_ok(int R, int N, int K) { return R >= 1 && R <= 16 * SK_MAXB && N >= 1 && K >= 1; }
static bool sk_al(const void *p, int ld) { return ((uintptr_t)p & 15) == 0 && ld % 4 == 0; }

// C (R x N, pitch ldc) = A (R x K, pitch lda) B (N x K, pitch ldb)^T (+ bias[N]); R <= 64, K % 4 == 0, A / B rows 16-byte aligned.
// act: 0 none, 1 ReLU, 2 LeakyReLU(0.01), applied after the bias.
extern "C" int pdgn_skinny_nt_act(int R, int N, int K, const float *A, int lda, const float *B, int ldb, const float *bias, float *C,
                                  int ldc, int act, pdgn_stream_t stream) {
    if (!sk_ok(R, N, K) || K % 4 || lda < K || ldb < K || ldc < N || !A || !B || !C || act < 0 || act > 2) return PDGN_ERR_INVALID;
    if (!sk_al(A, lda) || !sk_al(B, ldb)) return -2;
    hipStream_t s = (hipStream_t)stream;
    if (N < 4096)
        hipLaunchKernelGGL(skinny_nt_kernel<true>, dim3(cdiv(N, 16)), dim3(SK_THREADS), 0, s, R, N, K, A, lda, B, ldb, bias, C, ldc, act);
    else
        hipLaunchKernelGGL(skinny_nt_kernel<false>, dim3(cdiv(N, 64)), dim3(SK_THREADS), 0, s, R, N, K, A, lda, B, ldb, bias, C, ldc, act);
    return pdgn_launch_status();
}

extern "C" int pdgn_skinny_nt(int R, int N, int K, const float *A, int lda, const float *B, int ldb, const float *bias, float *C,
                              int ldc, pdgn_stream_t stream) {
    return pdgn_skinny_nt_act(R, N, K, A, lda, B, ldb, bias, C, ldc, 0, stream);
}

// C (R x N, pitch ldc) += A (R x K, pitch lda) B (K x N, pitch ldb): C is ZERO-FILLED by the caller (several K slices add into it
// with fp32 atomics); R <= 64, N % 4 == 0, B rows 16-byte aligned, A rows 16-byte aligned.
extern "C" int pdgn_skinny_nn(int R, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                              pdgn_stream_t stream) {
    if (!sk_ok(R, N, K) || N % 4 || lda < K || ldb < N || ldc < N || !A || !B || !C) return PDGN_ERR_INVALID;
    if (!sk_al(A, lda) || !sk_al(B, ldb)) return -2;
    const int kchunks = (K + 15) / 16, gx = cdiv(N, 64);
    int slices = 256 / gx;                                       // ~256 workgroups, at least 8 steps (two per wave) each: few
    slices = slices < 1 ? 1 : slices;                            // atomics per output element
    if (slices > kchunks / 8) slices = kchunks / 8;
    slices = slices < 1 ? 1 : slices;
    const int per = (kchunks + slices - 1) / slices;
    hipLaunchKernelGGL(skinny_nn_kernel<false>, dim3(gx, cdiv(kchunks, per)), dim3(SK_THREADS), 0, (hipStream_t)stream, R, N, K, per, A,
                       lda, B, ldb, C, ldc, nullptr, 0, 0);
    return pdgn_launch_status();
}

// The same with A[r][k] scaled by act'(P[r][k]) on load (act 1 = ReLU, 2 = LeakyReLU(0.01); P (R x K, pitch ldp) = the layer's
// pre-activation): dx = (dy * act'(pre)) W of a frozen Linear + activation.  C ZERO-FILLED by the caller.
extern "C" int pdgn_skinny_nn_masked(int R, int N, int K, const float *A, int lda, const float *P, int ldp, int act, const float *B,
                                     int ldb, float *C, int ldc, pdgn_stream_t stream) {
    if (!sk_ok(R, N, K) || N % 4 || lda < K || ldp < K || ldb < N || ldc < N || !A || !B || !C || !P || act < 1 || act > 2)
        return PDGN_ERR_INVALID;
    if (!sk_al(A, lda) || !sk_al(B, ldb)) return -2;
    const int kchunks = (K + 15) / 16, gx = cdiv(N, 64);
    int slices = 256 / gx;
    slices = slices < 1 ? 1 : slices;
    if (slices > kchunks / 8) slices = kchunks / 8;
    slices = slices < 1 ? 1 : slices;
    const int per = (kchunks + slices - 1) / slices;
    hipLaunchKernelGGL(skinny_nn_kernel<true>, dim3(gx, cdiv(kchunks, per)), dim3(SK_THREADS), 0, (hipStream_t)stream, R, N, K, per, A,
                       lda, B, ldb, C, ldc, P, ldp, act);
    return pdgn_launch_status();
}

// C (N x K, pitch ldc) = A (R x N, pitch lda)^T B (R x K, pitch ldb); R <= 64.
extern "C" int pdgn_skinny_tn(int R, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                              pdgn_stream_t stream) {
    if (!sk_ok(R, N, K) || lda < N || ldb < K || ldc < K || !A || !B || !C) return PDGN_ERR_INVALID;
    const int gx = cdiv(N, 64), kblocks = (K + 15) / 16;
    int gy = 512 / gx;
    gy = gy < 1 ? 1 : (gy > kblocks ? kblocks : gy);
    hipLaunchKernelGGL(skinny_tn_kernel, dim3(gx, gy), dim3(SK_THREADS), 0, (hipStream_t)stream, R, N, K, A, lda, B, ldb, C, ldc);
    return pdgn_launch_status();
}
