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
    const unsigned *max_a, *max_w;    // gemm_x3, two-part (fp16) instances: the maximum of |x| (bit pattern) of every ROW of each operand as the
                                      // kernel sees it (M entries / N entries: for an operand given transposed, of every column of the
                                      // matrix in memory); row r is multiplied by 2^e_r, e_r = x2_exponent(its maximum), before the split and
                                      // the result by 2^-e_A[row] 2^-e_W[column] (pre-split second operand: its rows' maxima sit behind
                                      // its two planes; the host points max_w there)
};

// Two-part mode: an operand's scales -- one power of two PER ROW of the operand as the kernel sees it (round 6; one per operand until
// then: a block-scaled format whose block was the whole matrix).  The maxima travel as bit patterns of |x| (unsigned order =
// magnitude order, so producers combine them with integer max / atomicMax in any order: deterministic); e = 14 - floor(log2 max):
// max |x| 2^e in [2^14, 2^15) -- nothing overflows fp16, and a value keeps 22 bits while it is within 2^-16 of ITS ROW's largest.
__device__ __forceinline__ int x2_exponent(unsigned maxbits) {     // (NaN / Inf: the largest exponent field; the products carry them)
    const int e = 14 - ((int)(maxbits >> 23) - 127);
    return e > 126 ? 126 : (e < -126 ? -126 : e);
}
__device__ __forceinline__ int x2_scale_field(unsigned maxbits) {  // exponent field of 2^e: 127 + e = 268 - field(max), clamped like x2_exponent
    const int f = 268 - (int)(maxbits >> 23);
    return f < 1 ? 1 : (f > 253 ? 253 : f);
}
__device__ __forceinline__ float x2_scale(unsigned maxbits) { return __int_as_float(x2_scale_field(maxbits) << 23); }          // 2^e
__device__ __forceinline__ float x2_unscale(unsigned maxbits) { return __int_as_float((254 - x2_scale_field(maxbits)) << 23); }  // 2^-e

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
