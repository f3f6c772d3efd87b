// gemm_x3.hip -- the dense contractions of the point-major layers: fp32 operands, products on the bf16 matrix cores.
//
//   C[m, n] = sum_k A[m, k] * W[n, k]  (+ bias[n]) (+ addend[m, n])        A (M x K), W (N x K) row-major, all fp32
//
// (models/PDGNet_v2.py:559-625, 835-862, 886-1014: every conv / linear of the deconvolution stack and of the discriminators
// once activations are point-major; with the weight transposed, their input gradients; with both, their weight gradients.)
//
// gfx950 multiplies fp32 on the matrix cores at 157 TFLOP/s (v_mfma_f32_16x16x4_f32: gemm_nt.hip runs at 85-88 % of that)
// and bf16 at 2.5 PFLOP/s.  Every fp32 operand value is split IN REGISTERS, after the fragment read from LDS, into three bf16
// parts by round-to-nearest conversions of successive remainders,
//       x = h + m + l (+ e),   |m| <= 2^-8 |x|,  |l| <= 2^-16 |x|,  |e| <= 2^-25 |x|  (less than half an fp32 ulp),
// and a product is the six partial products of weight <= 2^-16, each exact in the fp32 accumulator's input:
//       a w ~= al wh + am wm + ah wl + am wh + ah wm + ah wh          (v_mfma_f32_32x32x16_bf16, smallest terms first)
// The three dropped ones (am wl, al wm, al wl) are <= 2^-23 |a w| together: the error per product is about that of ONE
// rounded fp32 multiply (2^-24), and the accumulation is the matrix core's fp32 one, as in gemm_nt.hip.  Six bf16
// instructions of 32 cycles do the work of sixteen fp32 ones of 32 cycles (tools/split_mfma_probe.hip: 2.4 PFLOP/s of bf16
// MFMA = 400 TFLOP/s fp32-equivalent before the splitting).  tests/test_gpu_deconv.py compares both kernels with fp64.
//
// Everything around the products is gemm_nt.hip's: persistent workgroups walking (tile, k range) items with ONE continuous
// sequence of 32-deep k chunks through a two-stage LDS ring, operands global -> LDS by LDS-DMA in fp32 (the same swizzled
// [rows][32 floats] image), counted waits, XCD-aware tile order, data-parallel launch + stream-K tail with fp32 atomics,
// the epilogues (bias, addend, per-group row bias, LeakyReLU, gate, block-shifted BatchNorm partial sums).  What differs:
//   * a chunk is two 16-deep k steps of 32 x 32 fragments; lane (i, g) of a fragment holds k = 8g .. 8g + 7 of row i: two
//     ds_read_b128 (the 16-B columns 2g, 2g + 1 of the step, conflict-free in the swizzled image);
//   * the fragments of the NEXT k step are read and split (11 vector instructions per pair of values) in the issue slots the
//     running MFMAs leave free: a 32 x 32 x 16 bf16 MFMA holds the SIMD's vector issue for 8 of its 32 cycles
//     (MI355X_MICROARCH.md, cycle constants), so ~5 single-issue instructions per MFMA hide;
//   * D row (8q + 4g + r) = weight row, D column i = activation row: a lane holds four runs of 4 consecutive output
//     columns per 32 x 32 block: 16-B stores, issued inside the next item's first k step as in gemm_nt.hip.
#include <type_traits>

#include "gemm_shared.h"

#ifndef X3_ABLATE
#define X3_ABLATE 0                   // tools/x3_bench.hip (measurement only): 1 no splits, 2 no refills, 4 no stores, 8 no barrier, 16 no LDS reads
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct X3Parts {                      // eight values of a fragment lane as three bf16x8
    u32x4 h, m, l;
};

__device__ __forceinline__ unsigned x3_cvt_pk(float a, float b) {          // v_cvt_pk_bf16_f32: round to nearest even
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// two fp32 values -> dword e of each part
__device__ __forceinline__ void x3_split_pair(float a, float b, X3Parts &P, int e) {
    const unsigned h = x3_cvt_pk(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);      // exact
    const unsigned m = x3_cvt_pk(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);    // exact
    P.h[e] = h;
    P.m[e] = m;
    P.l[e] = x3_cvt_pk(sa, sb);
}

__device__ __forceinline__ f32x16 x3_mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// TM x TN blocks of 32 x 32 per wave, WM x WN waves, OCC workgroups per CU by registers.
template <int TM, int TN, int WM, int WN, int OCC, bool ATOMIC, bool WT, bool AT, bool EPI = false>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_x3_kernel(const NtArgs p) {
    constexpr int NW = WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN;
    // WT / AT: the operand is given transposed (K x N / K x M row-major), its chunk is [32 k][BN or BM] in LDS in 1-KB
    // pieces padded apart so that the b32 fragment reads of rows k and k + 8 (the two lane groups) fall on different banks.
    constexpr int WROWS = 256 / (BN > 256 ? 256 : BN), WPAD = WT ? 4 * WROWS : 0;
    constexpr int AROWS = 256 / (BM > 256 ? 256 : BM), APAD = AT ? 4 * AROWS : 0;
    static_assert(!AT || (BM <= 256 && 256 % BM == 0 && 8 % AROWS == 0), "A^T pieces: whole rows, a divisor of 8 per piece");
    static_assert(!WT || (BN <= 256 && 256 % BN == 0 && 8 % WROWS == 0), "W^T pieces: whole rows, a divisor of 8 per piece");
    constexpr int A_FLOATS = BM * NT_BK + (BM / 8) * APAD;
    constexpr int STAGE_FLOATS = A_FLOATS + BN * NT_BK + (BN / 8) * WPAD;
    constexpr int NPA = BM / 8 / NW, NPB = BN / 8 / NW, NP = NPA + NPB;   // 1-KB DMA pieces per wave and chunk
    constexpr int NS = ATOMIC ? 0 : 4 * TM * TN;                          // counted stores per wave and item
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "pieces must divide over the waves");
    constexpr int TOTAL = 6 * TM * TN;                                    // MFMAs of one k step
    constexpr int NPAIR = 4 * (TM + TN);                                  // value pairs to split per k step
    static_assert(TOTAL >= 2 * NP, "a k step must have room for the DMA pieces between its MFMAs");
    __shared__ __attribute__((aligned(1024))) float smem[2 * STAGE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lg = lane >> 5;

    // XCD-aware decode of the 1-D grid: ids are dealt round-robin to the 8 XCDs; XCD x takes a contiguous range
    const int G = gridDim.x, pid = blockIdx.x;
    const int gq = G >> 3, gr = G & 7, xcd = pid & 7;
    const int v = xcd * gq + min(xcd, gr) + (pid >> 3);

    const int KC = p.kchunks;
    // tile index -> (tile row, tile column), grouped: NT_GROUP_M tile rows are walked column by column (gemm_nt.hip)
    auto decode = [&](int tile, int &tm, int &tn) {
        const int per_group = NT_GROUP_M * p.tiles_n;
        const int grp = tile / per_group, r = tile - grp * per_group;
        const int first = grp * NT_GROUP_M;
        const int gsz = min(NT_GROUP_M, p.tiles_m - first);
        tn = r / gsz;
        tm = first + (r - tn * gsz);
    };
    struct Cur {
        int tile, kb, kc, ke;       // current item: tile, first / next / end chunk
        int j;                      // DP: next item index
        long long f, f1;            // SK: next flattened position, end
        bool valid;
    };
    auto next_item = [&](Cur &c) {
        if (ATOMIC) {
            c.valid = c.f < c.f1;
            if (c.valid) {
                c.tile = p.tile_begin + (int)(c.f / KC);
                c.kc = c.kb = (int)(c.f % KC);
                const long long left = c.f1 - c.f;
                c.ke = (left < (long long)(KC - c.kc)) ? c.kc + (int)left : KC;
                c.f += c.ke - c.kc;
            }
        } else {
            c.tile = p.tile_begin + v + c.j * G;
            c.valid = c.tile < p.tile_end;
            c.kc = c.kb = 0;
            c.ke = KC;
            c.j++;
        }
    };
    Cur ld, cp;
    ld.j = 0;
    ld.f = (long long)v * p.sk_per_wg;
    {
        const long long total = (long long)(p.tile_end - p.tile_begin) * KC;
        ld.f1 = min(total, ld.f + p.sk_per_wg);
    }
    ld.tile = ld.kb = ld.kc = ld.ke = 0;
    ld.valid = false;
    next_item(ld);
    cp = ld;
    if (!ld.valid) return;

    // ---- per-lane constants of the DMA: piece i of a wave covers rows 8*(wave + i*NW) .. +7 of the A (then W) part
    const int drow = lane >> 3;                                   // row inside the piece == (row & 7)
    const int dcol = (lane & 7) ^ drow;                           // 16-B source column stored at position lane & 7
    unsigned voffA[NPA], voffB[NPB];
#pragma unroll
    for (int i = 0; i < NPA; ++i)
        voffA[i] = AT ? (unsigned)(((wave + i * NW) * AROWS + lane / (BM / 4)) * p.lda + (lane % (BM / 4)) * 4) * 4u
                      : (unsigned)(((wave + i * NW) * 8 + drow) * p.lda + dcol * 4) * 4u;
#pragma unroll
    for (int i = 0; i < NPB; ++i)
        voffB[i] = WT ? (unsigned)(((wave + i * NW) * WROWS + lane / (BN / 4)) * p.ldw + (lane % (BN / 4)) * 4) * 4u
                      : (unsigned)(((wave + i * NW) * 8 + drow) * p.ldw + dcol * 4) * 4u;

    i32x4 rsA, rsW;
    long long ld_m0 = 0;
    int ld_n0 = 0, ld_mrows = 0, ld_nrows = 0;
    auto make_srds = [&](int tile) {
        int tm, tn;
        decode(tile, tm, tn);
        ld_m0 = (long long)tm * BM;
        ld_n0 = tn * BN;
        ld_mrows = (int)min((long long)BM, p.M - ld_m0);
        ld_nrows = min(BN, p.N - ld_n0);
        if (!AT) rsA = nt_srd(p.A + ld_m0 * p.lda, (unsigned)((long long)ld_mrows * p.lda * 4));
        if (!WT) rsW = nt_srd(p.W + (long long)ld_n0 * p.ldw, (unsigned)(ld_nrows * p.ldw * 4));
    };
    make_srds(ld.tile);

    const unsigned smem_base = (unsigned)(size_t)(lds_void_t *)smem;
    int ld_k0 = 0;
    bool ld_kok = true;
    unsigned ld_dst = 0, ld_dst_w = 0, ld_dst_a = 0;
    auto issue_begin = [&](int stage) {
        ld_k0 = ld.kc * NT_BK;
        ld_kok = ld_k0 + dcol * 4 < p.K;                           // K % 4 == 0: a 16-B column is all in or all out
        ld_dst = smem_base + (unsigned)(stage * STAGE_FLOATS + wave * 256) * 4u;
        ld_dst_w = smem_base + (unsigned)(stage * STAGE_FLOATS + A_FLOATS + wave * (256 + WPAD)) * 4u;
        ld_dst_a = smem_base + (unsigned)(stage * STAGE_FLOATS + wave * (256 + APAD)) * 4u;
        const int krows = min(NT_BK, p.K - ld_k0);                 // reduction rows of this chunk: the rest reads as zero
        if (AT) rsA = nt_srd(p.A + (long long)ld_k0 * p.lda + ld_m0, (unsigned)(((long long)(krows - 1) * p.lda + ld_mrows) * 4));
        if (WT) rsW = nt_srd(p.W + (long long)ld_k0 * p.ldw + ld_n0, (unsigned)(((long long)(krows - 1) * p.ldw + ld_nrows) * 4));
    };
    auto issue_piece = [&](int i) {
        if (i < NPA) {
            const int j = i < NPA ? i : 0;
            if (!AT) nt_dma16(rsA, ld_dst + j * NW * 1024, ld_kok ? voffA[j] : NT_OOB, ld_k0 * 4);
            else nt_dma16(rsA, ld_dst_a + j * NW * (1024 + APAD * 4), voffA[j], 0);
        } else {
            const int j = i < NPA ? 0 : i - NPA;
            if (!WT) nt_dma16(rsW, ld_dst + A_FLOATS * 4 + j * NW * 1024, ld_kok ? voffB[j] : NT_OOB, ld_k0 * 4);
            else nt_dma16(rsW, ld_dst_w + j * NW * (1024 + WPAD * 4), voffB[j], 0);
        }
    };
    auto advance_load = [&]() {
        ld.kc++;
        if (ld.kc == ld.ke) {
            next_item(ld);
            if (ld.valid) make_srds(ld.tile);
        }
    };

    // ---- fragment reads.  Row-major operand: row li of a 32-row block, the 16-B columns 4s + 2 lg and + 1 of k step s,
    // stored at (column ^ (row & 7)): the second one is the first's position ^ 1.
    const int fo[2] = {li * NT_BK + 4 * ((2 * lg) ^ (li & 7)), li * NT_BK + 4 * ((4 + 2 * lg) ^ (li & 7))};
    const int abase = wm * 32 * TM * NT_BK, bbase = A_FLOATS + wn * 32 * TN * NT_BK;
    // transposed operand: k row r of the chunk sits at float (r / ROWS) * (256 + PAD) + (r % ROWS) * B; a lane's element u
    // of k step s is row 16 s + 8 lg + u (8 lg never carries across a piece: ROWS | 8), column w * 32 T + 32 block + li
    const int wt_lane = ((8 * lg) / WROWS) * (256 + WPAD) + wn * 32 * TN + li;
    const int at_lane = ((8 * lg) / AROWS) * (256 + APAD) + wm * 32 * TM + li;
    float rawa[TM][8], raww[TN][8];                                // the k step being split
    auto read_a = [&](int stage, int s, int a) {
        if (!AT) {
            const float *sa = smem + stage * STAGE_FLOATS + abase + a * 32 * NT_BK;
            const float4 x = *reinterpret_cast<const float4 *>(sa + fo[s]);
            const float4 y = *reinterpret_cast<const float4 *>(sa + (fo[s] ^ 4));
            rawa[a][0] = x.x; rawa[a][1] = x.y; rawa[a][2] = x.z; rawa[a][3] = x.w;
            rawa[a][4] = y.x; rawa[a][5] = y.y; rawa[a][6] = y.z; rawa[a][7] = y.w;
        } else {
            const float *sa = smem + stage * STAGE_FLOATS + at_lane + 32 * a;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = 16 * s + u;
                rawa[a][u] = sa[(r / AROWS) * (256 + APAD) + (r % AROWS) * BM];
            }
        }
    };
    auto read_w = [&](int stage, int s, int b) {
        if (!WT) {
            const float *sb = smem + stage * STAGE_FLOATS + bbase + b * 32 * NT_BK;
            const float4 x = *reinterpret_cast<const float4 *>(sb + fo[s]);
            const float4 y = *reinterpret_cast<const float4 *>(sb + (fo[s] ^ 4));
            raww[b][0] = x.x; raww[b][1] = x.y; raww[b][2] = x.z; raww[b][3] = x.w;
            raww[b][4] = y.x; raww[b][5] = y.y; raww[b][6] = y.z; raww[b][7] = y.w;
        } else {
            const float *sb = smem + stage * STAGE_FLOATS + A_FLOATS + wt_lane + 32 * b;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = 16 * s + u;
                raww[b][u] = sb[(r / WROWS) * (256 + WPAD) + (r % WROWS) * BN];
            }
        }
    };
    auto read_step = [&](int stage, int s) {
#pragma unroll
        for (int a = 0; a < TM; ++a) read_a(stage, s, a);
#pragma unroll
        for (int b = 0; b < TN; ++b) read_w(stage, s, b);
    };

    X3Parts pa[2][TM], pw[2][TN];                                  // [parity of the k step]
    // pair j of a k step (A fragments first): split into the parts of parity `par`
    auto split_pair = [&](int par, int j) {
        const int f = j >> 2, e = j & 3;
        if (f < TM) x3_split_pair(rawa[f][2 * e], rawa[f][2 * e + 1], pa[par][f], e);
        else x3_split_pair(raww[f - TM][2 * e], raww[f - TM][2 * e + 1], pw[par][f - TM], e);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- output of a finished item.  Sub-block bb = 4 b + q of a lane's accumulators: output row 32 a + li, output columns
    // 32 b + 8 q + 4 lg + (0 .. 3).  The 16-B stores of item t are issued from inside the first k step of item t + 1 (before
    // the MFMA that restarts the block from zero), in the shadow of running MFMAs.
    constexpr int NBB = 4 * TN;
    const int mloc0 = wm * 32 * TM + li, nloc0 = wn * 32 * TN + 4 * lg;
    auto coloff = [](int bb) { return 32 * (bb >> 2) + 8 * (bb & 3); };
    auto get4 = [&](int a, int bb) {
        const int b = bb >> 2, q = bb & 3;
        return (f32x4){acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
    };
    auto set4 = [&](int a, int bb, const f32x4 x) {
        const int b = bb >> 2, q = bb & 3;
        acc[a][b][4 * q] = x[0]; acc[a][b][4 * q + 1] = x[1]; acc[a][b][4 * q + 2] = x[2]; acc[a][b][4 * q + 3] = x[3];
    };
    __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)p.C, 0, 0, 0x00020000);   // nothing pending: all out of range
    unsigned st_off[NBB];                                          // byte offset of (row mloc0, sub-block bb), or out of range
#pragma unroll
    for (int bb = 0; bb < NBB; ++bb) st_off[bb] = NT_OOB;
    auto store_block = [&](int a, int b) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned off = st_off[4 * b + q] + (unsigned)(a * 32 * p.ldc) * 4u;
            if (!(X3_ABLATE & 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, get4(a, 4 * b + q)), rsC, off, 0, 0);
        }
    };
    auto finish_item = [&]() {
        int tm, tn;
        decode(cp.tile, tm, tn);
        const long long m0 = (long long)tm * BM;
        const int n0 = tn * BN;
        const long long mrows = min((long long)BM, p.M - m0);
        const int ncols = min(BN, p.N - n0);
        if (!ATOMIC) {
            rsC = __builtin_amdgcn_make_buffer_rsrc((void *)(p.C + m0 * p.ldc + n0), 0, (int)(mrows * p.ldc * 4), 0x00020000);
#pragma unroll
            for (int bb = 0; bb < NBB; ++bb)
                st_off[bb] = (nloc0 + coloff(bb) < ncols && !(p.dbg & 1)) ? (unsigned)(mloc0 * p.ldc + nloc0 + coloff(bb)) * 4u : NT_OOB;
            if (p.addend) {                                        // out of range reads 0
                __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)(p.addend + m0 * p.ldadd + n0), 0,
                                                                               (int)(mrows * p.ldadd * 4), 0x00020000);
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb)
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const int ml = mloc0 + 32 * a, nl = nloc0 + coloff(bb);
                        set4(a, bb, get4(a, bb) + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rsD, nl < ncols ? (unsigned)(ml * p.ldadd + nl) * 4u : NT_OOB, 0, 0)));
                    }
            }
            if (p.bias) {
                __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)(p.bias + n0), 0, ncols * 4, 0x00020000);
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb) {
                    const f32x4 bz = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(nloc0 + coloff(bb)) * 4u, 0, 0));
#pragma unroll
                    for (int a = 0; a < TM; ++a) set4(a, bb, get4(a, bb) + bz);
                }
            }
            if (EPI && p.row_bias) {                               // a bias per group of rows: the heads' per-sample term
                __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)p.row_bias, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const unsigned row = (unsigned)(m0 + mloc0 + 32 * a);
                    const unsigned grp = row < (unsigned)p.M ? (p.rows_per_group == 1 ? row : __umulhi(row, p.rpg_magic)) : 0u;
#pragma unroll
                    for (int bb = 0; bb < NBB; ++bb) {
                        const int nl = nloc0 + coloff(bb);
                        set4(a, bb, get4(a, bb) + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rsR, nl < ncols ? (grp * (unsigned)p.ld_rb + n0 + nl) * 4u : NT_OOB, 0, 0)));
                    }
                }
            }
            if (EPI && p.act == 2) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][b][r] = acc[a][b][r] > 0.f ? acc[a][b][r] : 0.01f * acc[a][b][r];
            }
            if (EPI && p.gate) {
                __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void *)(p.gate + m0 * p.ldgate + n0), 0,
                                                                               (int)(mrows * p.ldgate * 4), 0x00020000);
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb)
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const int ml = mloc0 + 32 * a, nl = nloc0 + coloff(bb);
                        const f32x4 g = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                     rsG, nl < ncols ? (unsigned)(ml * p.ldgate + nl) * 4u : NT_OOB, 0, 0));
                        f32x4 x = get4(a, bb);
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[r] *= g[r] > 0.f ? 1.f : 0.01f;
                        set4(a, bb, x);
                    }
            }
            if (p.stat_part) {
                // per-column statistics of the wave's 32 TM rows, shifted by the block's first row (pv): sum (x - pv),
                // sum (x - pv)^2 and pv (gemm_nt.hip; combined in fp64 by cl_finalize_blocks_kernel).  In registers over a,
                // then over the 32 lanes of equal lg; one partial row of [3N] floats per (tile row, wave row).
                float *P = p.stat_part + ((size_t)tm * WM + wm) * 3 * p.N;
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb) {
                    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f}, pv[4];
                    const f32x4 x0 = get4(0, bb);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = __shfl(x0[r], lane & 32, 64);            // row li = 0 of block a = 0
#pragma unroll
                    for (int a = 0; a < TM; ++a)
                        if (mloc0 + 32 * a < mrows) {
                            const f32x4 x = get4(a, bb);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float d = x[r] - pv[r];
                                cs[r] += d;
                                cq[r] = __fmaf_rn(d, d, cq[r]);
                            }
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int d = 1; d < 32; d <<= 1) {
                            cs[r] += __shfl_xor(cs[r], d, 64);
                            cq[r] += __shfl_xor(cq[r], d, 64);
                        }
                    const int nl = nloc0 + coloff(bb);
                    if (li == 0 && nl < ncols) {
                        *reinterpret_cast<float4 *>(P + n0 + nl) = make_float4(cs[0], cs[1], cs[2], cs[3]);
                        *reinterpret_cast<float4 *>(P + p.N + n0 + nl) = make_float4(cq[0], cq[1], cq[2], cq[3]);
                        *reinterpret_cast<float4 *>(P + 2 * p.N + n0 + nl) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                    }
                }
            }
        } else {
            // partial tile (operands swapped): D row (8q + 4 lg + r) = activation row, D column li = weight row: 128-B
            // atomic segments.  The holder of the tile's first chunk adds the bias / addend.
            const bool head = cp.kb == 0;
            const int mla = wm * 32 * TM + 4 * lg, nla = wn * 32 * TN + li;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int nl = nla + 32 * b;
                const bool nok = nl < ncols;
                const float bz = (head && p.bias && nok) ? p.bias[n0 + nl] : 0.f;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int ml = mla + 32 * a + 8 * (j >> 2) + (j & 3);
                        if (nok && ml < mrows) {
                            float o = acc[a][b][j] + bz;
                            if (head && p.addend) o += p.addend[(m0 + ml) * p.ldadd + n0 + nl];
                            atomicAdd(p.C + (m0 + ml) * p.ldc + n0 + nl, o);
                        }
                    }
            }
        }
    };

    // ---- one k step: the 6 TM TN MFMAs on the parts of parity CUR; between them (fixed places, fenced against the
    // compiler's scheduler) the split of the step read before into parity CUR ^ 1 (SPLIT) and the DMA pieces of the refill
    // (DMA).  FIRST: the step restarts the accumulators, each block's pending stores right before its restart.
    auto k_step = [&](auto cur_c, auto first_c, auto split_c, auto dma_c) {
        constexpr int CUR = decltype(cur_c)::value;
        constexpr bool FIRST = decltype(first_c)::value, SPLIT = decltype(split_c)::value, DMA = decltype(dma_c)::value;
        constexpr int LEAD = 2;                                    // MFMAs before the first split: the reads are landing
        constexpr int EVERY = TOTAL / NP;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                f32x16 c = acc[a][b];
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const int idx = (a * TN + b) * 6 + t;
                    if (FIRST && t == 0) {
                        if (!ATOMIC) {
                            __builtin_amdgcn_sched_barrier(0);
                            store_block(a, b);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) c[r] = 0.f;
                    }
                    const X3Parts &A = pa[CUR][a], &W = pw[CUR][b];
                    const u32x4 ap = t == 0 ? A.l : (t == 1 || t == 3) ? A.m : A.h;          // al wh, am wm, ah wl, am wh, ah wm, ah wh
                    const u32x4 wp = t == 2 ? W.l : (t == 1 || t == 4) ? W.m : W.h;
                    c = ATOMIC ? x3_mfma(ap, wp, c) : x3_mfma(wp, ap, c);
                    if (SPLIT) {
                        // pairs [idx' * NPAIR / (TOTAL - LEAD), ...) after MFMA idx = LEAD + idx'
                        const int i0 = idx - LEAD;
                        if (i0 >= 0) {
                            const int j0 = (i0 * NPAIR) / (TOTAL - LEAD), j1 = ((i0 + 1) * NPAIR) / (TOTAL - LEAD);
                            if (j1 > j0) {
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = j0; j < j1; ++j)
                                    if (!(X3_ABLATE & 1)) split_pair(CUR ^ 1, j);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    if (DMA && idx % EVERY == EVERY / 2 && idx / EVERY < NP) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (!(X3_ABLATE & 2)) issue_piece(idx / EVERY);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                acc[a][b] = c;
            }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::true_type T;
    typedef std::false_type F;

    // ---- prologue: chunks 0 and 1 in flight, chunk 0 landed, its first k step split
    long long todo;                                                // chunks of this workgroup's sequence still to multiply
    if (ATOMIC) todo = ld.f1 - (long long)v * p.sk_per_wg;
    else todo = (long long)((p.tile_end - 1 - (p.tile_begin + v)) / G + 1) * KC;
    issue_begin(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_piece(i);
    advance_load();
    if (ld.valid) {
        issue_begin(1);
#pragma unroll
        for (int i = 0; i < NP; ++i) issue_piece(i);
        advance_load();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_step(0, 0);
#pragma unroll
    for (int j = 0; j < NPAIR; ++j) split_pair(0, j);
    int stage = 0;
    while (cp.valid) {
        bool first = true;
        for (;;) {
            // (A) the chunk's second k step read and split while its first is multiplied
            if (!(X3_ABLATE & 16)) read_step(stage, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (first) k_step(I0(), T(), T(), F());
            else k_step(I0(), F(), T(), F());
            // (B) everyone has read this stage to the end, and the next chunk has landed everywhere: my own pieces
            // (counted wait: only the stores of (A) were issued after them), then the barrier
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 63 ? 63 : NS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(X3_ABLATE & 8)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // (C) the next chunk's first k step read and split, this stage refilled with the chunk after next, while the
            // second k step is multiplied
            first = false;
            todo--;
            if (todo > 0 && !(X3_ABLATE & 16)) read_step(stage ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ld.valid) {
                issue_begin(stage);
                k_step(I1(), F(), T(), T());
                advance_load();
            } else {
                k_step(I1(), F(), T(), F());
            }
            stage ^= 1;
            cp.kc++;
            if (cp.kc == cp.ke) break;
        }
        __builtin_amdgcn_sched_barrier(0);
        finish_item();
        __builtin_amdgcn_sched_barrier(0);
        next_item(cp);
    }
    // the last item's result
    if (!ATOMIC) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) store_block(a, b);
    }
}

// ------------------------------------------------------------------ host side
template <int TM, int TN, int WM, int WN, int OCC>
struct X3Cfg {
    static constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    static constexpr int LDS = 2 * ((BM + BN) * NT_BK + (BM / 8 + BN / 8) * 16) * 4;        // with both operands' piece padding
    static constexpr int WG_PER_CU = OCC;

    struct Plan {
        int tiles_m, tiles_n, kchunks, grid_dp, dp_tiles, grid_sk;
        long long sk_per_wg;
        double cost;                                               // launch model, microseconds
    };
    // Launch model as in gemm_nt.hip, with this loop's rate: a CU multiplies ~1.0 Mflop/us (fp32-equivalent).
    static int dp_grid(long long tiles, int slots, int kc) {
        int tpw = (64 + kc - 1) / kc;
        tpw = tpw < 1 ? 1 : (tpw > 8 ? 8 : tpw);
        const long long g = (tiles + tpw - 1) / tpw;
        const long long lo = tiles < slots ? tiles : slots;
        return (int)(g < lo ? lo : g);
    }
    static double chunk_us(int w) { return 2.0 * BM * BN * NT_BK * w / x3_rate(); }
    static double x3_rate() {
        static double r = 0.0;
        if (r == 0.0) { const char *e = getenv("PDGN_X3_RATE"); r = e ? atof(e) * 1e6 : 1.0e6; }
        return r;
    }
    static Plan plan(long long m, int n, int k, bool allow_sk) {
        Plan pl;
        pl.tiles_m = cdiv(m, BM);
        pl.tiles_n = cdiv(n, BN);
        pl.kchunks = cdiv(k, NT_BK);
        const long long T = (long long)pl.tiles_m * pl.tiles_n;
        const int cus = nt_cus(), slots = cus * WG_PER_CU, KC = pl.kchunks;
        const long long rounds = T / slots, tail = T - rounds * slots;
        const double ov = 3.0;
        const double full = (double)rounds * (KC + ov) * chunk_us(WG_PER_CU);
        pl.dp_tiles = (int)T;
        pl.grid_dp = dp_grid(T, slots, KC);
        pl.grid_sk = 0;
        pl.sk_per_wg = 0;
        pl.cost = full + (tail ? (KC + ov) * chunk_us((int)((tail + cus - 1) / cus)) : 0.0);
        if (allow_sk && tail > 0) {
            const long long iters = tail * KC;
            const long long cand[3] = {(long long)slots, (long long)cus, iters / 16};
            for (int ci = 0; ci < 3; ++ci) {
                long long g = cand[ci] < 1 ? 1 : (cand[ci] > slots ? slots : cand[ci]);
                g = g < iters ? g : iters;
                const long long per = (iters + g - 1) / g;
                if (per < 8 && g > 1) continue;                    // short ranges: the atomics cost more than they balance
                g = (iters + per - 1) / per;
                const double zero_rows = (double)(m - (long long)((rounds * slots) / (NT_GROUP_M * pl.tiles_n)) * NT_GROUP_M * BM);
                const double c = full + (per + ov) * chunk_us((int)((g + cus - 1) / cus)) + (double)(g + tail) * BM * BN * 4 / 1.3e6 +
                                 zero_rows * n * 4 / 3.0e6 + 4.0;
                if (c < pl.cost) {
                    pl.cost = c;
                    pl.dp_tiles = (int)(rounds * slots);
                    pl.grid_dp = rounds ? dp_grid(rounds * slots, slots, KC) : 0;
                    pl.grid_sk = (int)g;
                    pl.sk_per_wg = per;
                }
            }
        }
        return pl;
    }

    template <bool ATOMIC, bool WT, bool AT, bool EPI>
    static void go(int grid, hipStream_t s, const NtArgs &a) {
        hipLaunchKernelGGL((gemm_x3_kernel<TM, TN, WM, WN, OCC, ATOMIC, WT, AT, EPI>), dim3(grid), dim3(64 * WM * WN), 0, s, a);
    }

    template <bool WT, bool AT = false>
    static int launch(long long m, int n, int k, const float *A, int lda, const float *W, int ldw, const float *bias,
                      const float *addend, int ldadd, float *C, int ldc, float *stat_part, hipStream_t s,
                      const NtEpi &epi = NtEpi()) {
        const bool allow_sk = stat_part == nullptr && ldc == n && !epi.any();
        const Plan pl = plan(m, n, k, allow_sk);
        NtArgs a;
        a.M = m; a.N = n; a.K = k; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldadd = ldadd;
        a.A = A; a.W = W; a.bias = bias; a.addend = addend; a.C = C; a.stat_part = stat_part;
        a.row_bias = epi.row_bias; a.ld_rb = epi.ld_rb; a.rows_per_group = epi.rows_per_group > 0 ? epi.rows_per_group : 1;
        a.rpg_magic = a.rows_per_group > 1 ? (unsigned)(0x100000000ULL / (unsigned)a.rows_per_group) + 1u : 0u;
        a.act = epi.act; a.gate = epi.gate; a.ldgate = epi.ldgate;
        a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n; a.kchunks = pl.kchunks;
        { const char *e = getenv("PDGN_NT_DBG"); a.dbg = e ? atoi(e) : 0; }
        const long long T = (long long)pl.tiles_m * pl.tiles_n;
        if (pl.grid_sk) {
            const long long r0 = (long long)(pl.dp_tiles / (NT_GROUP_M * pl.tiles_n)) * NT_GROUP_M * BM;
            if (hipMemsetAsync(C + r0 * ldc, 0, (size_t)(m - r0) * ldc * sizeof(float), s) != hipSuccess) return pdgn_launch_status();
        }
        if (pl.grid_dp) {
            a.tile_begin = 0; a.tile_end = pl.dp_tiles; a.sk_per_wg = 0;
            if (!AT && epi.any()) go<false, WT, false, true>(pl.grid_dp, s, a);
            else go<false, WT, AT, false>(pl.grid_dp, s, a);
        }
        if (pl.grid_sk) {
            a.tile_begin = pl.dp_tiles; a.tile_end = (int)T; a.sk_per_wg = pl.sk_per_wg;
            go<true, WT, AT, false>(pl.grid_sk, s, a);
        }
        return pdgn_launch_status();
    }
};

typedef X3Cfg<2, 2, 2, 2, 1> X3Square;   // 128 x 128, 4 waves of 64 x 64 (64 KB): one per CU (the parts of two k steps + accumulators exceed 256 registers)
typedef X3Cfg<4, 2, 2, 2, 1> X3Big;      // 256 x 128, 4 waves of 128 x 64 (96 KB): one per CU, one wave per SIMD
typedef X3Cfg<2, 1, 2, 2, 2> X3Narrow;   // 128 x 64, 4 waves of 64 x 32 (48 KB): two per CU

static int x3_mode() {                   // PDGN_GEMM: "x3" (default) or "fp32" (gemm_nt.hip: the fp32 matrix instructions)
    const char *e = getenv("PDGN_GEMM");
    return (e && e[0] == 'f') ? 0 : 1;
}

static int x3_pick(long long m, int n, int k, bool stats) {
    const char *e = getenv("PDGN_NT_CFG");          // measurement / tests only: 0 .. 2 (3, the fp32 kernel's fourth, reads as 2)
    if (e && *e) return atoi(e) < 0 ? 0 : (atoi(e) > 2 ? 2 : atoi(e));
    const bool sk = !stats;
    const double c[3] = {X3Big::plan(m, n, k, sk).cost, X3Square::plan(m, n, k, sk).cost, X3Narrow::plan(m, n, k, sk).cost};
    int best = 1;
    for (int i = 0; i < 3; ++i)
        if (c[i] < c[best] * 0.97) best = i;
    return best;
}

static bool x3_args_ok(long long m, int n, int k, int lda, int ldw, int ldadd, int ldc, const float *addend, bool wt) {
    return m >= 1 && n >= 4 && k >= 4 && n % 4 == 0 && k % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && ldc % 4 == 0 &&
           lda >= k && ldw >= (wt ? n : k) && ldc >= n && (!addend || (ldadd % 4 == 0 && ldadd >= n)) && lda < (1 << 19) &&
           ldw < (1 << 19) && ldc < (1 << 19) && (long long)cdiv(m, 128) * cdiv(n, 64) < 0x7fffffffLL &&
           (!wt || (long long)k * ldw < (1LL << 27));
}

template <bool WT>
static int x3_dispatch(long long m, int n, int k, const float *A, int lda, const float *W, int ldw, const float *bias,
                       const float *addend, int ldadd, float *C, int ldc, float *stat_part, hipStream_t s,
                       const NtEpi &epi = NtEpi()) {
    switch (x3_pick(m, n, k, stat_part != nullptr || epi.any())) {
        case 0: return X3Big::launch<WT>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, s, epi);
        case 2: return X3Narrow::launch<WT>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, s, epi);
        default: return X3Square::launch<WT>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, s, epi);
    }
}

// C (m x n, row pitch ldc) = A (m x k, pitch lda) W (n x k, pitch ldw)^T (+ bias[n]) (+ addend (m x n, pitch ldadd)).
// n, k and every pitch are multiples of 4 floats and all base pointers 16-byte aligned.  stat_part (may be NULL):
// pdgn_gemm_nt_stat_rows(m, n, k) rows of [3n] floats = per-column sum (x - pv) | sum (x - pv)^2 | pv of row blocks of C
// (pv: the block's first row; pdgn_gemm_nt_stat_block_rows rows per block) for pdgn_bn_stats_from_gemm_partials.
extern "C" int pdgn_gemm_nt(long long m, int n, int k, const float *A, int lda, const float *W, int ldw,
                            const float *bias, const float *addend, int ldadd, float *C, int ldc, float *stat_part,
                            pdgn_stream_t stream) {
    if (!x3_mode()) return fp32_gemm_nt(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, stream);
    if (!x3_args_ok(m, n, k, lda, ldw, ldadd, ldc, addend, false)) return PDGN_ERR_INVALID;
    return x3_dispatch<false>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream);
}

// The same product with the second operand given transposed: C (m x n) = A (m x k) Wt (k x n, row pitch ldw) -- the
// input gradient dX = dY W of a dense layer straight from its (C_out x C_in) weight.
extern "C" int pdgn_gemm_nn(long long m, int n, int k, const float *A, int lda, const float *Wt, int ldw,
                            const float *bias, const float *addend, int ldadd, float *C, int ldc, float *stat_part,
                            pdgn_stream_t stream) {
    if (!x3_mode()) return fp32_gemm_nn(m, n, k, A, lda, Wt, ldw, bias, addend, ldadd, C, ldc, stat_part, stream);
    if (!x3_args_ok(m, n, k, lda, ldw, ldadd, ldc, addend, true)) return PDGN_ERR_INVALID;
    return x3_dispatch<true>(m, n, k, A, lda, Wt, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream);
}

// pdgn_gemm_nt / pdgn_gemm_nn (transposed_w != 0) with the extended epilogue:  C = act(A W^T + bias + addend +
// row_bias[row / rows_per_group]) * lrelu'(gate)  -- a bias per group of rows (the per-sample term of the generator's heads,
// models/PDGNet_v2.py:835-862 on cat([g broadcast, x])), LeakyReLU(0.01) on the result (act = 2), and / or the LeakyReLU
// derivative of a saved activation as a factor (gate: the result is the gradient wrt that layer's PRE-activation).  Each of
// them replaces a full elementwise pass over C.  No stream-K tail (whole tiles only), like a launch with statistics.
extern "C" int pdgn_gemm_nt_ex(long long m, int n, int k, const float *A, int lda, const float *W, int ldw, const float *bias,
                               const float *addend, int ldadd, float *C, int ldc, float *stat_part, const float *row_bias,
                               int ld_rb, int rows_per_group, int act, const float *gate, int ldgate, int transposed_w,
                               pdgn_stream_t stream) {
    if (!x3_mode())
        return fp32_gemm_nt_ex(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, row_bias, ld_rb, rows_per_group,
                               act, gate, ldgate, transposed_w, stream);
    if (!x3_args_ok(m, n, k, lda, ldw, ldadd, ldc, addend, transposed_w != 0)) return PDGN_ERR_INVALID;
    if ((act != 0 && act != 2) || (row_bias && (ld_rb < n || ld_rb % 4 || rows_per_group < 1)) ||
        (gate && (ldgate < n || ldgate % 4)) || m >= (1LL << 31) || (row_bias && m * rows_per_group >= (1LL << 32)))
        return PDGN_ERR_INVALID;
    NtEpi e;
    e.row_bias = row_bias; e.ld_rb = ld_rb; e.rows_per_group = rows_per_group; e.act = act; e.gate = gate; e.ldgate = ldgate;
    return transposed_w ? x3_dispatch<true>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream, e)
                        : x3_dispatch<false>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream, e);
}

// Weight gradient of a point-major dense layer, dW (n x k) = dY (m x n)^T X (m x k): the same kernel with BOTH operands
// given transposed, the reduction over the m rows split over the workgroups (stream-K launch, fp32 atomics into dW,
// which the launch zero-fills itself).  For outputs of at least one 128 x 64 tile; pdgn_gemm_tn (gemm_tn.hip) keeps the
// small ones.
extern "C" int pdgn_gemm_tn_big(long long m, int n, int k, const float *dY, int ldy, const float *X, int ldx, float *dW,
                                pdgn_stream_t stream) {
    if (!x3_mode()) return fp32_gemm_tn_big(m, n, k, dY, ldy, X, ldx, dW, stream);
    if (m < 1 || n < 4 || k < 4 || n % 4 || k % 4 || ldy % 4 || ldx % 4 || ldy < n || ldx < k || ldy >= (1 << 19) ||
        ldx >= (1 << 19) || m > 0x7fffffffLL * 16)
        return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    // kernel roles: output rows = n (columns of dY), output columns = k (columns of X), reduction = m
    switch (x3_pick(n, k, (int)(m > 0x7fffffff ? 0x7fffffff : m), false)) {
        case 0: return X3Big::launch<true, true>(n, k, (int)m, dY, ldy, X, ldx, nullptr, nullptr, 0, dW, k, nullptr, s);
        case 2: return X3Narrow::launch<true, true>(n, k, (int)m, dY, ldy, X, ldx, nullptr, nullptr, 0, dW, k, nullptr, s);
        default: return X3Square::launch<true, true>(n, k, (int)m, dY, ldy, X, ldx, nullptr, nullptr, 0, dW, k, nullptr, s);
    }
}

// Number of [3n] partial-statistics rows pdgn_gemm_nt writes for this problem (tile rows x waves along m).
extern "C" long long pdgn_gemm_nt_stat_rows(long long m, int n, int k) {
    if (!x3_mode()) return fp32_gemm_nt_stat_rows(m, n, k);
    if (m < 1 || n < 1 || k < 1) return PDGN_ERR_INVALID;
    switch (x3_pick(m, n, k, true)) {
        case 0: return (long long)cdiv(m, X3Big::BM) * 2;
        case 2: return (long long)cdiv(m, X3Narrow::BM) * 2;
        default: return (long long)cdiv(m, X3Square::BM) * 2;
    }
}

// Rows of C each of those partial rows covers (partial p: rows p * block .. ; the blocks past m are empty).
extern "C" int pdgn_gemm_nt_stat_block_rows(long long m, int n, int k) {
    if (!x3_mode()) return fp32_gemm_nt_stat_block_rows(m, n, k);
    if (m < 1 || n < 1 || k < 1) return PDGN_ERR_INVALID;
    switch (x3_pick(m, n, k, true)) {
        case 0: return X3Big::BM / 2;
        case 2: return X3Narrow::BM / 2;
        default: return X3Square::BM / 2;
    }
}

// Tile configuration pdgn_gemm_nt picks for a problem, + 16 when a stream-K launch follows the data-parallel one; host-side
// only.  x3: 0: 256x128, 1: 128x128, 2: 128x64;  fp32: 0: 256x128, 1: 128x128, 2: 160x64, 3: 128x64.
extern "C" int pdgn_gemm_nt_config(long long m, int n, int k, int with_stats) {
    if (!x3_mode()) return fp32_gemm_nt_config(m, n, k, with_stats);
    if (m < 1 || n < 1 || k < 1) return PDGN_ERR_INVALID;
    const int c = x3_pick(m, n, k, with_stats != 0);
    const bool sk = !with_stats;
    const int g = c == 0 ? X3Big::plan(m, n, k, sk).grid_sk : c == 1 ? X3Square::plan(m, n, k, sk).grid_sk
                                                                      : X3Narrow::plan(m, n, k, sk).grid_sk;
    return c + (g ? 16 : 0);
}
