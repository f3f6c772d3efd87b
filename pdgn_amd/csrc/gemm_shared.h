// gemm_shared.h -- what the two dense-contraction kernels (gemm_nt.hip: fp32 matrix instructions; gemm_x3.hip: fp32 operands
// split into three bf16 parts) have in common: operand descriptors, the LDS-DMA instruction, the argument block.
#pragma once
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

#define NT_BK 32
#define NT_GROUP_M 8
#define NT_OOB 0x40000000u            // a byte offset past every tile descriptor (num_records < 2^30)

typedef int i32x4 __attribute__((ext_vector_type(4)));

// Buffer descriptor (raw, stride 0): offsets >= bytes read as zero / are not written.
__device__ __forceinline__ i32x4 nt_srd(const void *base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// One LDS-DMA wave instruction: 64 lanes x 16 B, lane l -> LDS byte lds_base + 16 l, from descriptor offset
// voff (per lane) + soff (scalar).  Issued as inline asm so that the compiler does not order every later LDS read
// behind it with s_waitcnt vmcnt(0) -- the kernel counts these operations itself.
__device__ __forceinline__ void nt_dma16(i32x4 srd, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_base), "v"(voff), "s"(srd), "s"(soff)
                 : "memory", "m0");
}

struct NtArgs {
    long long M;
    int N, K, lda, ldw, ldc, ldadd;
    const float *A, *W, *bias, *addend;
    float *C, *stat_part;
    int tiles_m, tiles_n, tile_begin, tile_end, kchunks;
    long long sk_per_wg;              // stream-K launch: (tile, chunk) iterations per workgroup
    int sk_split;                     // > 0 (gemm_x3 only): split-K launch over sk_split tiles: workgroup v takes tile v % sk_split,
                                      // chunks [v / sk_split * sk_per_wg, + sk_per_wg) -- workgroups that run together share panels
    float *sk_ws;                     // gemm_x3, non-atomic instances: partial tiles (whole, pitch BN) go to this workspace instead of
                                      // into C (stream-K tail without atomics).  sk_split > 0: workgroup v's one tile at slot v;
    int sk_seg;                       // sk_seg > 0 (sk_split == 0): the FLATTENED order -- workgroup v takes (tile, chunk) iterations
                                      // [v sk_per_wg, + sk_per_wg), its i-th tile's part at slot v * sk_seg + i
    int dbg;                          // builds with -DPDGN_NT_DEBUG only (ablation: 1 = stores dropped); always 0 otherwise
    // extended epilogue (pdgn_gemm_nt_ex), applied in this order after bias / addend:
    const float *row_bias;            // + row_bias[(row / rows_per_group) * ld_rb + col]: a bias per GROUP of rows (per sample)
    int ld_rb, rows_per_group;
    unsigned rpg_magic;               // row / rows_per_group == umulhi(row, rpg_magic) (rows_per_group > 1)
    int rb_bytes;                     // extent of the row_bias table in bytes (its buffer descriptor's range, < NT_OOB)
    int act;                          // 2: LeakyReLU(0.01) on the result
    const float *gate;                // result *= (gate[row, col] > 0 ? 1 : 0.01): the LeakyReLU derivative of a saved activation
    int ldgate;
    const unsigned short *Wp;         // gemm_x3, PW instances: the second operand pre-split into three bf16 planes [N][ldw]
    long long wplane;                 // elements between the planes
    const unsigned *max_a, *max_w;    // gemm_x3, two-part (fp16) instances: X2_PARTS partial maxima (bit patterns of |x|) of each operand,
                                      // which is multiplied by 2^e, e = x2_exponent(their maximum), before the split (pre-split
                                      // second operand: its exponent sits behind its two planes instead)
};

// Two-part mode: an operand's scale.  A scan (x2_absmax_kernel, gemm_x3.hip) leaves X2_PARTS partial maxima of |x| (as bit patterns:
// unsigned order = magnitude order; unused entries zero) in a 1-KB slot; every consumer reduces them itself (one load per thread of
// a 256-thread workgroup) -- no atomics, no fences, nothing to re-arm.  e = 14 - floor(log2 max): max |x| 2^e in [2^14, 2^15).
#define X2_PARTS 256
__device__ __forceinline__ int x2_exponent(unsigned maxbits) {     // (NaN / Inf: the largest exponent field; the products carry them)
    const int e = 14 - ((int)(maxbits >> 23) - 127);
    return e > 126 ? 126 : (e < -126 ? -126 : e);
}
// max over the workgroup's threads (a multiple of 64, at most 512) of two values each; scratch: 16 words of LDS; every thread must
// call (two barriers)
__device__ __forceinline__ void x2_block_max2(unsigned &a, unsigned &b, unsigned *scratch) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a = max(a, (unsigned)__shfl_xor((int)a, o));
        b = max(b, (unsigned)__shfl_xor((int)b, o));
    }
    const int nw = (int)(blockDim.x >> 6);
    if ((threadIdx.x & 63) == 0) {
        scratch[threadIdx.x >> 6] = a;
        scratch[8 + (threadIdx.x >> 6)] = b;
    }
    __syncthreads();
    a = scratch[0];
    b = scratch[8];
    for (int i = 1; i < nw; ++i) {
        a = max(a, scratch[i]);
        b = max(b, scratch[8 + i]);
    }
    __syncthreads();
}

// ------------------------------------------------------------------ host side
struct NtDev {
    int cus;
};
static inline int nt_cus() {
    static int cached = 0;
    if (!cached) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            return 256;
        cached = cus;
    }
    return cached;
}

// Process-wide switches of the dense contractions.  Read from the environment ONCE (first use); tests and tools change them
// through pdgn_gemm_set_mode / pdgn_gemm_set_config (no getenv per launch, no environment mutation at run time).
//   mode: 2 = products on the matrix cores, two scaled fp16 parts per value where that pays and three bf16 parts elsewhere (gemm_x3.hip,
//   default), 1 = three bf16 parts everywhere (PDGN_GEMM=x3), 0 = fp32 matrix instructions (gemm_nt.hip; PDGN_GEMM=fp32)
//   cfg:  -1 = the launch model's pick (default), 0 .. 3 = a forced tile configuration (PDGN_NT_CFG; measurement / tests)
//   shape16: per instance class of gemm_x3.hip, which bf16 matrix instruction it runs on (PDGN_X3_SHAPE / PDGN_X3_SHAPE16_MASK;
//   pdgn_gemm_set_shape)
struct NtSwitches {
    int mode, cfg, splitk, shape16, shape16_default;
};
NtSwitches &nt_switches();

struct NtEpi {                        // extended epilogue of pdgn_gemm_nt_ex (all optional)
    const float *row_bias = nullptr;
    int ld_rb = 0, rows_per_group = 1, act = 0;
    const float *gate = nullptr;
    int ldgate = 0;
    bool any() const { return row_bias || act || gate; }
};


// The fp32-matrix-instruction forms of the public entry points (gemm_nt.hip); gemm_x3.hip dispatches between the two.
int fp32_gemm_nt(long long m, int n, int k, const float *A, int lda, const float *W, int ldw, const float *bias,
                 const float *addend, int ldadd, float *C, int ldc, float *stat_part, pdgn_stream_t stream);
int fp32_gemm_nn(long long m, int n, int k, const float *A, int lda, const float *Wt, int ldw, const float *bias,
                 const float *addend, int ldadd, float *C, int ldc, float *stat_part, pdgn_stream_t stream);
int fp32_gemm_nt_ex(long long m, int n, int k, const float *A, int lda, const float *W, int ldw, const float *bias,
                    const float *addend, int ldadd, float *C, int ldc, float *stat_part, const float *row_bias, int ld_rb,
                    int rows_per_group, int act, const float *gate, int ldgate, int transposed_w, pdgn_stream_t stream);
int fp32_gemm_tn_big(long long m, int n, int k, const float *dY, int ldy, const float *X, int ldx, float *dW,
                     pdgn_stream_t stream);
long long fp32_gemm_nt_stat_rows(long long m, int n, int k);
int fp32_gemm_nt_stat_block_rows(long long m, int n, int k);
int fp32_gemm_nt_config(long long m, int n, int k, int with_stats);
