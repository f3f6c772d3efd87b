// thin.hip -- dense layers with at most 4 channels on one side: the xyz-in layers (Conv1d(3, 64) of the
// discriminators, conv_xyz of the deconvolution: models/PDGNet_v2.py:559-566, 886-1014) and the xyz-out layer of the
// heads (Conv1d(64, 3), :835-862), forward, input gradient and weight gradient.  These are streaming passes over the
// wide operand (a (rows, 64) matrix read or written once), not matrix-core work: on the MFMA kernel they cost a
// zero-padded copy of the 3-wide operand per call (fill + copy launches on the issuing thread's critical path).
//   thin_k:  Y (m, n)     = X (m, k<=4)  W'^T (+ bias)     n % 4 == 0      [+ BatchNorm partial sums of Y]
//   thin_n:  Y (m, n<=4)  = X (m, k)     W'^T (+ bias)     k % 4 == 0
//   thin_tn: O (ta, wb)   = A (m, ta<=4)^T B (m, wb)       wb % 4 == 0     [+ column sums of A, of B]
// W'[j, kk] sits at W[j * wrs + kk * wcs], O[i, j] at O[i * osi + j * osj]: the same kernels serve a weight and its
// transpose (input gradients), and both orientations of a weight gradient.
#include "common.h"

#define THIN_THREADS 256
#define THIN_ROWS 256             // rows per workgroup of thin_k (one partial-statistics row each)
#define THIN_TN_ROWS 512          // rows per workgroup of thin_tn

__global__ __launch_bounds__(THIN_THREADS) void thin_k_kernel(long long m, int n, int k, const float *__restrict__ X, int ldx,
                                                              const float *__restrict__ W, int wrs, int wcs,
                                                              const float *__restrict__ bias, float *__restrict__ Y, int ldy,
                                                              float *__restrict__ part, const float *__restrict__ gate, int ldg) {
    __shared__ float red[THIN_THREADS * 8];
    const int n4 = n >> 2;                       // threads per row
    const int rpp = THIN_THREADS / n4;           // rows per pass
    const int c4 = threadIdx.x % n4, ro = threadIdx.x / n4;
    float w[4][4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b[j] = bias ? bias[c4 * 4 + j] : 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w[j][kk] = kk < k ? W[(size_t)(c4 * 4 + j) * wrs + (size_t)kk * wcs] : 0.f;
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f};
    const long long r0 = (long long)blockIdx.x * THIN_ROWS;
    const long long r1 = r0 + THIN_ROWS < m ? r0 + THIN_ROWS : m;
    if (part) {
        // the statistics are sums of (y - pv), pv = this block's first output row (recomputed by every thread with the
        // row loop's own expression): shifted sums do not cancel when |mean| >> std (see cl_finalize_blocks_kernel)
        const float *xr = X + r0 * ldx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = b[j];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __fmaf_rn(kk < k ? xr[kk] : 0.f, w[j][kk], acc);
            pv[j] = acc;
        }
    }
    if (ro < rpp) {
        for (long long r = r0 + ro; r < r1; r += rpp) {
            const float *xr = X + r * ldx;
            float x[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) x[kk] = kk < k ? xr[kk] : 0.f;
            float4 o;
            float *ov = &o.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = b[j];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __fmaf_rn(x[kk], w[j][kk], acc);
                ov[j] = acc;
                if (gate) ov[j] = acc = acc * (gate[r * ldg + c4 * 4 + j] > 0.f ? 1.f : 0.01f);   // LeakyReLU derivative of a saved activation
                const float d = acc - pv[j];
                s1[j] += d;
                s2[j] = __fmaf_rn(d, d, s2[j]);
            }
            *reinterpret_cast<float4 *>(Y + r * ldy + c4 * 4) = o;
        }
    }
    if (!part) return;
    // per-column sum (y - pv) | sum (y - pv)^2 | pv of this workgroup's rows: one partial row of [3n] floats
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[threadIdx.x * 8 + j] = s1[j]; red[threadIdx.x * 8 + 4 + j] = s2[j]; }
    __syncthreads();
    for (int t = threadIdx.x; t < n4 * 8; t += THIN_THREADS) {
        const int cc = t >> 3, j = t & 7;                             // column group, component (4 sums | 4 squares)
        float acc = 0.f;
        for (int q = 0; q < rpp; ++q) acc += red[(q * n4 + cc) * 8 + j];
        part[(size_t)blockIdx.x * 3 * n + (j < 4 ? 0 : n) + cc * 4 + (j & 3)] = acc;
    }
    if (ro == 0)
        *reinterpret_cast<float4 *>(part + (size_t)blockIdx.x * 3 * n + 2 * n + c4 * 4) = make_float4(pv[0], pv[1], pv[2], pv[3]);
}

// v + (v of the lane `CTRL` names inside the 16-lane DPP row; 0 where there is none): one v_add_f32 with a DPP operand
template <int CTRL>
__device__ __forceinline__ float thin_row_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// 16 lanes per row (= one DPP row), each a float4 of the row per step; the n <= 4 dot products are summed across the 16 lanes by
// four row_shr adds each (the total lands in the row's last lane) -- as ds_bpermute butterflies they were 16 LDS-crossbar
// operations per thread, more than the row's own load.  VECW: W has unit column stride and 16-byte aligned rows (float4 loads).
template <bool VECW>
__global__ __launch_bounds__(THIN_THREADS) void thin_n_kernel(long long m, int n, int k, const float *__restrict__ X, int ldx,
                                                              const float *__restrict__ W, int wrs, int wcs,
                                                              const float *__restrict__ bias, float *__restrict__ Y, int ldy) {
    const int l = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * (THIN_THREADS / 16) + (threadIdx.x >> 4);
    const bool live = row < m;
    const float *xr = X + (live ? row : 0) * ldx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k4 = l; k4 < (k >> 2); k4 += 16) {
        const float4 xv = *reinterpret_cast<const float4 *>(xr + k4 * 4);
        const float *x = &xv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < n) {
                if (VECW) {
                    const float4 wv = *reinterpret_cast<const float4 *>(W + (size_t)j * wrs + k4 * 4);
                    acc[j] = __fmaf_rn(x[0], wv.x, acc[j]);
                    acc[j] = __fmaf_rn(x[1], wv.y, acc[j]);
                    acc[j] = __fmaf_rn(x[2], wv.z, acc[j]);
                    acc[j] = __fmaf_rn(x[3], wv.w, acc[j]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j] = __fmaf_rn(x[e], W[(size_t)j * wrs + (size_t)(k4 * 4 + e) * wcs], acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc[j] = thin_row_add<0x111>(acc[j]);                      // row_shr 1, 2, 4, 8: lane 15 of the row holds the sum
        acc[j] = thin_row_add<0x112>(acc[j]);
        acc[j] = thin_row_add<0x114>(acc[j]);
        acc[j] = thin_row_add<0x118>(acc[j]);
    }
    if (live && l == 15) {
        float *y = Y + row * ldy;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n) y[j] = acc[j] + (bias ? bias[j] : 0.f);
    }
}

// O[i, j] += sum_r A[r, i] B[r, j] over this workgroup's rows (O zero-filled by the caller); optionally the column sums
// of A (sum_a[ta]) and of B (sum_b[wb]) -- the bias gradient of the layer, whichever operand is its dY.
__global__ __launch_bounds__(THIN_THREADS) void thin_tn_kernel(long long m, int ta, int wb, const float *__restrict__ A, int lda,
                                                               const float *__restrict__ B, int ldb, float *__restrict__ O,
                                                               int osi, int osj, float *__restrict__ sum_a,
                                                               float *__restrict__ sum_b) {
    __shared__ float red[THIN_THREADS * 20];
    const int w4 = wb >> 2;
    const int rpp = THIN_THREADS / w4;
    const int c4 = threadIdx.x % w4, ro = threadIdx.x / w4;
    float acc[5][4];                              // rows 0..3: A column i times B; row 4: column sums of B
    float sa[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    const long long r0 = (long long)blockIdx.x * THIN_TN_ROWS;
    const long long r1 = r0 + THIN_TN_ROWS < m ? r0 + THIN_TN_ROWS : m;
    if (ro < rpp) {
        for (long long r = r0 + ro; r < r1; r += rpp) {
            const float4 bv = *reinterpret_cast<const float4 *>(B + r * ldb + c4 * 4);
            const float *b = &bv.x;
            const float *ar = A + r * lda;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = i < ta ? ar[i] : 0.f;
                sa[i] += a;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] = __fmaf_rn(a, b[e], acc[i][e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4][e] += b[e];
        }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[threadIdx.x * 20 + i * 4 + e] = acc[i][e];
    __syncthreads();
    // thread t < w4 * 20 sums one (column group, component) over the row offsets
    for (int t = threadIdx.x; t < w4 * 20; t += THIN_THREADS) {
        const int cc = t / 20, comp = t % 20, i = comp >> 2, e = comp & 3;
        float s = 0.f;
        for (int q = 0; q < rpp; ++q) s += red[(q * w4 + cc) * 20 + comp];
        const int j = cc * 4 + e;
        if (i < 4) {
            if (i < ta) atomicAdd(O + (size_t)i * osi + (size_t)j * osj, s);
        } else if (sum_b) {
            atomicAdd(sum_b + j, s);
        }
    }
    if (sum_a) {
        __syncthreads();
        // every thread of column group 0 saw all of its rows' A values: reduce those over the row offsets
        if (c4 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[ro * 4 + i] = sa[i];
        }
        __syncthreads();
        if (threadIdx.x < ta) {
            float s = 0.f;
            for (int q = 0; q < rpp; ++q) s += red[q * 4 + threadIdx.x];
            atomicAdd(sum_a + threadIdx.x, s);
        }
    }
}

static bool thin_aligned(const void *p, int ld) { return ((uintptr_t)p & 15) == 0 && ld % 4 == 0; }

extern "C" long long pdgn_thin_stat_rows(long long m) { return (m + THIN_ROWS - 1) / THIN_ROWS; }
extern "C" int pdgn_thin_stat_block_rows(void) { return THIN_ROWS; }      // rows of Y per [3n] partial row

// gate (k <= 4 form only, may be NULL; (m x n), pitch ldgate): Y *= (gate > 0 ? 1 : 0.01) -- the LeakyReLU derivative of a
// saved activation, when Y is the gradient wrt that activation (the heads' backward, no separate elementwise pass).
extern "C" int pdgn_thin_nt_ex(long long m, int n, int k, const float *X, int ldx, const float *W, int wrs, int wcs,
                               const float *bias, float *Y, int ldy, float *stat_part, const float *gate, int ldgate,
                               pdgn_stream_t stream) {
    if (m <= 0 || n <= 0 || k <= 0 || !X || !W || !Y) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (k <= 4 && n % 4 == 0 && n <= 4 * THIN_THREADS) {
        if (!thin_aligned(Y, ldy) || (gate && ldgate < n)) return -2;
        hipLaunchKernelGGL(thin_k_kernel, dim3(cdiv(m, THIN_ROWS)), dim3(THIN_THREADS), 0, s, m, n, k, X, ldx, W, wrs, wcs, bias,
                           Y, ldy, stat_part, gate, ldgate);
        return pdgn_launch_status();
    }
    if (gate) return -3;
    if (n <= 4 && k % 4 == 0 && !stat_part) {
        if (!thin_aligned(X, ldx)) return -2;
        if (wcs == 1 && wrs % 4 == 0 && ((uintptr_t)W & 15) == 0)
            hipLaunchKernelGGL(thin_n_kernel<true>, dim3(cdiv(m, THIN_THREADS / 16)), dim3(THIN_THREADS), 0, s, m, n, k, X, ldx, W, wrs,
                               wcs, bias, Y, ldy);
        else
            hipLaunchKernelGGL(thin_n_kernel<false>, dim3(cdiv(m, THIN_THREADS / 16)), dim3(THIN_THREADS), 0, s, m, n, k, X, ldx, W, wrs,
                               wcs, bias, Y, ldy);
        return pdgn_launch_status();
    }
    return -3;
}

extern "C" int pdgn_thin_nt(long long m, int n, int k, const float *X, int ldx, const float *W, int wrs, int wcs,
                            const float *bias, float *Y, int ldy, float *stat_part, pdgn_stream_t stream) {
    return pdgn_thin_nt_ex(m, n, k, X, ldx, W, wrs, wcs, bias, Y, ldy, stat_part, nullptr, 0, stream);
}

extern "C" int pdgn_thin_tn(long long m, int ta, int wb, const float *A, int lda, const float *B, int ldb, float *O, int osi,
                            int osj, float *sum_a, float *sum_b, pdgn_stream_t stream) {
    if (m <= 0 || ta <= 0 || ta > 4 || wb <= 0 || wb % 4 || wb > 4 * THIN_THREADS || !A || !B || !O) return -1;
    if (!thin_aligned(B, ldb)) return -2;
    hipLaunchKernelGGL(thin_tn_kernel, dim3(cdiv(m, THIN_TN_ROWS)), dim3(THIN_THREADS), 0, (hipStream_t)stream, m, ta, wb, A, lda, B,
                       ldb, O, osi, osj, sum_a, sum_b);
    return pdgn_launch_status();
}
