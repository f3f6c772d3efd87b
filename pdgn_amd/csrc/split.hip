// split.hip -- an fp32 matrix as three bf16 planes (x = h + m + l), once per weight.
//
// gemm_x3.hip forms every fp32 product from the operands' three-way bf16 splits and normally splits both operands in its
// loaders, once per workgroup and k chunk.  A WEIGHT matrix is the same for every row tile of a launch, for both generator
// passes of an iteration and for the input-gradient product of the backward pass: pdgn_split_bf16x3 writes its parts once --
// with exactly the loader's arithmetic (round-to-nearest bf16 of the value, of the exact remainder, of the second exact
// remainder: gemm_x3.hip::x3_split_pair) -- as planes [rows][ld] for the forward form (pdgn_gemm_nt_ps) and, optionally,
// as planes of the TRANSPOSE [cols][ldt] (the input gradient dX = dY W is the same NT product against W^T).
// (No reference counterpart: models/PDGNet_v2.py's layers run on cuDNN / cuBLAS.)
#include "common.h"

typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned sp_cvt_pk(float a, float b) {          // v_cvt_pk_bf16_f32: round to nearest even
    const sp_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sp_bf16x2));
}

__device__ __forceinline__ void sp_split(float a, unsigned short &h, unsigned short &m, unsigned short &l) {
    const unsigned hh = sp_cvt_pk(a, 0.f);
    const float ra = a - __uint_as_float(hh << 16);                        // exact
    const unsigned mm = sp_cvt_pk(ra, 0.f);
    const float sa = ra - __uint_as_float(mm << 16);                       // exact
    h = (unsigned short)(hh & 0xffffu);
    m = (unsigned short)(mm & 0xffffu);
    l = (unsigned short)(sp_cvt_pk(sa, 0.f) & 0xffffu);
}

// 32 x 32 tiles, 256 threads (32 x 8): row-major planes straight from the registers, transposed planes through LDS.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(int rows, int cols, const float *__restrict__ src, int lds_,
                                                           unsigned short *__restrict__ P, int ldp, long long pstride,
                                                           unsigned short *__restrict__ PT, int ldpt, long long ptstride) {
    __shared__ unsigned short tile[3][32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        unsigned short h = 0, m = 0, l = 0;
        if (r < rows && c < cols) sp_split(src[(size_t)r * lds_ + c], h, m, l);
        if (P && r < rows && c < ldp) {                                    // the pad columns [cols, ld) hold zeros: the contraction's
            P[(size_t)r * ldp + c] = h;                                    // 16-B loads reach them on a K tail (0 x NaN would poison)
            P[pstride + (size_t)r * ldp + c] = m;
            P[2 * pstride + (size_t)r * ldp + c] = l;
        }
        tile[0][ty + 8 * i][tx] = h;
        tile[1][ty + 8 * i][tx] = m;
        tile[2][ty + 8 * i][tx] = l;
    }
    if (!PT) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;                        // output row = source column
        if (c < cols && r < ldpt) {                                        // (rows .. ldpt: zeros, as above)
            PT[(size_t)c * ldpt + r] = tile[0][tx][ty + 8 * i];
            PT[ptstride + (size_t)c * ldpt + r] = tile[1][tx][ty + 8 * i];
            PT[2 * ptstride + (size_t)c * ldpt + r] = tile[2][tx][ty + 8 * i];
        }
    }
}

// src (rows x cols, row pitch ld_src floats) -> planes (3 x [rows][ld_planes] bf16, plane_stride elements apart; may be NULL)
// and / or planes_t (3 x [cols][ld_planes_t], plane_stride_t apart; may be NULL): the parts h | m | l of every value.
extern "C" int pdgn_split_bf16x3(int rows, int cols, const float *src, int ld_src, unsigned short *planes, int ld_planes,
                                 long long plane_stride, unsigned short *planes_t, int ld_planes_t, long long plane_stride_t,
                                 pdgn_stream_t stream) {
    if (rows < 1 || cols < 1 || ld_src < cols || (!planes && !planes_t)) return PDGN_ERR_INVALID;
    if (planes && (ld_planes < cols || plane_stride < (long long)(rows - 1) * ld_planes + cols)) return PDGN_ERR_INVALID;
    if (planes_t && (ld_planes_t < rows || plane_stride_t < (long long)(cols - 1) * ld_planes_t + rows)) return PDGN_ERR_INVALID;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, rows, cols, src,
                       ld_src, planes, ld_planes, plane_stride, planes_t, ld_planes_t, plane_stride_t);
    return pdgn_launch_status();
}


// ---- two scaled fp16 parts (gemm_x3.hip, two-part mode): row r of the planes holds x 2^e_r = h + l, e_r = 14 - floor(log2 max |x|
// over that row) (gemm_shared.h x2_scale); planes [2][rows][ld] fp16 (h | l) and the rows' maxima (bit patterns of |x|, uint32[rows])
// right behind them (element offset 2 plane_stride: the contraction reads its second operand's scales from there).  The transposed
// planes' rows are the matrix's COLUMNS: scaled by the column maxima, which sit behind THEM.  Exactly the loader's arithmetic
// (gemm_x3.hip conv_pair).
typedef _Float16 sp_f16x2 __attribute__((ext_vector_type(2)));
#include "gemm_shared.h"
int x2_maxima_launch_ext(const float *X, long long rows, int cols, int ld, unsigned *rowmax, unsigned *colmax, hipStream_t s);      // gemm_x3.hip

__device__ __forceinline__ void sp_split2(float a, float sc, unsigned short &h, unsigned short &l) {
    const float a2 = a * sc;
    const sp_f32x2 v = {a2, 0.f};
    const sp_f16x2 hh = __builtin_convertvector(v, sp_f16x2);
    const sp_f32x2 r = {a2 - (float)hh[0], 0.f};
    const sp_f16x2 ll = __builtin_convertvector(r, sp_f16x2);
    h = (unsigned short)(__builtin_bit_cast(unsigned, hh) & 0xffffu);
    l = (unsigned short)(__builtin_bit_cast(unsigned, ll) & 0xffffu);
}

__global__ __launch_bounds__(256) void split_f16x2_kernel(int rows, int cols, const float *__restrict__ src, int lds_,
                                                          unsigned short *__restrict__ P, int ldp, long long pstride,
                                                          unsigned short *__restrict__ PT, int ldpt, long long ptstride) {
    __shared__ unsigned short tile[2][32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const unsigned *__restrict__ rmax = P ? reinterpret_cast<const unsigned *>(P + 2 * pstride) : nullptr;      // [rows], written by the scan in front
    const unsigned *__restrict__ cmax = PT ? reinterpret_cast<const unsigned *>(PT + 2 * ptstride) : nullptr;   // [cols]
    const int c = c0 + tx;
    const float scc = (PT && c < cols) ? x2_scale(cmax[c]) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i;
        const bool in = r < rows && c < cols;
        const float v = in ? src[(size_t)r * lds_ + c] : 0.f;
        unsigned short h = 0, l = 0;
        if (P && in) sp_split2(v, x2_scale(rmax[r]), h, l);
        if (P && r < rows && c < ldp) {                            // (pad columns [cols, ld) hold zeros: a K tail must not multiply 0 by garbage)
            P[(size_t)r * ldp + c] = h;
            P[pstride + (size_t)r * ldp + c] = l;
        }
        unsigned short ht = 0, lt = 0;
        if (PT && in) sp_split2(v, scc, ht, lt);
        tile[0][ty + 8 * i][tx] = ht;
        tile[1][ty + 8 * i][tx] = lt;
    }
    if (!PT) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = c0 + ty + 8 * i, r = r0 + tx;
        if (cc < cols && r < ldpt) {
            PT[(size_t)cc * ldpt + r] = tile[0][tx][ty + 8 * i];
            PT[ptstride + (size_t)cc * ldpt + r] = tile[1][tx][ty + 8 * i];
        }
    }
}

// src (rows x cols, pitch ld_src; cols and ld_src multiples of 4, 16-byte aligned) -> planes (2 x [rows][ld_planes] fp16,
// plane_stride elements apart, + the rows' maxima as uint32[rows] at element 2 plane_stride: the buffer holds 2 plane_stride + 2
// rows elements; plane_stride >= rows * ld_planes and a multiple of 8) and / or the same of the transpose (its rows = the
// columns: 2 plane_stride_t + 2 cols elements).
extern "C" int pdgn_split_f16x2(int rows, int cols, const float *src, int ld_src, unsigned short *planes, int ld_planes,
                                long long plane_stride, unsigned short *planes_t, int ld_planes_t, long long plane_stride_t,
                                pdgn_stream_t stream) {
    if (rows < 1 || cols < 1 || ld_src < cols || (!planes && !planes_t)) return PDGN_ERR_INVALID;
    // (the kernel fills the pad columns of every row, the last one's too: the planes must hold whole rows, and the
    // maxima behind them start on a 16-byte boundary)
    if (planes && (ld_planes < cols || plane_stride < (long long)rows * ld_planes || plane_stride % 8)) return PDGN_ERR_INVALID;
    if (planes_t && (ld_planes_t < rows || plane_stride_t < (long long)cols * ld_planes_t || plane_stride_t % 8)) return PDGN_ERR_INVALID;
    const int st = x2_maxima_launch_ext(src, rows, cols, ld_src, planes ? reinterpret_cast<unsigned *>(planes + 2 * plane_stride) : nullptr,
                                        planes_t ? reinterpret_cast<unsigned *>(planes_t + 2 * plane_stride_t) : nullptr, (hipStream_t)stream);
    if (st != 0) return st;
    hipLaunchKernelGGL(split_f16x2_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, rows, cols, src,
                       ld_src, planes, ld_planes, plane_stride, planes_t, ld_planes_t, plane_stride_t);
    return pdgn_launch_status();
}
