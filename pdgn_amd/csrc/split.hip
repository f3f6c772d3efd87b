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


// ---- two scaled fp16 parts (gemm_x3.hip, two-part mode): x 2^e = h + l, e = 14 - floor(log2 max |x|) from a scan of the matrix;
// planes [2][rows][ld] fp16 (h | l) and the exponent as ONE int32 right behind them (element offset 2 plane_stride: the
// contraction reads it from there), the same for the transpose.  Exactly the loader's arithmetic (gemm_x3.hip conv_pair).
typedef _Float16 sp_f16x2 __attribute__((ext_vector_type(2)));
#include "gemm_shared.h"
const unsigned *x2_scan(const float *X, long long rows, int cols, int ld, hipStream_t s);      // gemm_x3.hip

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
                                                          const unsigned *__restrict__ maxima, unsigned short *__restrict__ P, int ldp,
                                                          long long pstride, unsigned short *__restrict__ PT, int ldpt,
                                                          long long ptstride) {
    __shared__ unsigned short tile[2][32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    __shared__ unsigned red[16];
    unsigned mx = maxima[threadIdx.x], unused = 0;                // X2_PARTS = 256 = the workgroup
    x2_block_max2(mx, unused, red);
    const int e = x2_exponent(mx);
    const float sc = __int_as_float((127 + e) << 23);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (P) *reinterpret_cast<int *>(P + 2 * pstride) = e;
        if (PT) *reinterpret_cast<int *>(PT + 2 * ptstride) = e;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        unsigned short h = 0, l = 0;
        if (r < rows && c < cols) sp_split2(src[(size_t)r * lds_ + c], sc, h, l);
        if (P && r < rows && c < ldp) {
            P[(size_t)r * ldp + c] = h;
            P[pstride + (size_t)r * ldp + c] = l;
        }
        tile[0][ty + 8 * i][tx] = h;
        tile[1][ty + 8 * i][tx] = l;
    }
    if (!PT) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < ldpt) {
            PT[(size_t)c * ldpt + r] = tile[0][tx][ty + 8 * i];
            PT[ptstride + (size_t)c * ldpt + r] = tile[1][tx][ty + 8 * i];
        }
    }
}

// src (rows x cols, pitch ld_src) -> planes (2 x [rows][ld_planes] fp16, plane_stride elements apart, + one int32 at element
// 2 plane_stride: the buffer holds 2 plane_stride + 2 elements; plane_stride a multiple of 8) and / or the same of the transpose.
extern "C" int pdgn_split_f16x2(int rows, int cols, const float *src, int ld_src, unsigned short *planes, int ld_planes,
                                long long plane_stride, unsigned short *planes_t, int ld_planes_t, long long plane_stride_t,
                                pdgn_stream_t stream) {
    if (rows < 1 || cols < 1 || ld_src < cols || (!planes && !planes_t)) return PDGN_ERR_INVALID;
    if (planes && (ld_planes < cols || plane_stride < (long long)(rows - 1) * ld_planes + cols || plane_stride % 2)) return PDGN_ERR_INVALID;
    if (planes_t && (ld_planes_t < rows || plane_stride_t < (long long)(cols - 1) * ld_planes_t + rows || plane_stride_t % 2)) return PDGN_ERR_INVALID;
    const unsigned *e = x2_scan(src, rows, cols, ld_src, (hipStream_t)stream);
    if (!e) return PDGN_ERR_INVALID;                             // (no scale slots: pdgn_gemm_set_scale_slots)
    hipLaunchKernelGGL(split_f16x2_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, rows, cols, src,
                       ld_src, e, planes, ld_planes, plane_stride, planes_t, ld_planes_t, plane_stride_t);
    return pdgn_launch_status();
}
