// gemm_x3.hip -- the dense contractions of the point-major layers: fp32 operands, products on the bf16 matrix cores.
//
//   C[m, n] = sum_k A[m, k] * W[n, k]  (+ bias[n]) (+ addend[m, n])        A (M x K), W (N x K) row-major, all fp32
//
// (models/PDGNet_v2.py:559-625, 835-862, 886-1014: every conv / linear of the deconvolution stack and of the discriminators
// once activations are point-major; with the weight transposed, their input gradients; with both, their weight gradients.)
//
// gfx950 multiplies fp32 on the matrix cores at 157 TFLOP/s (v_mfma_f32_16x16x4_f32: gemm_nt.hip runs at 85-88 % of that)
// and bf16 at 2.5 PFLOP/s.  Every fp32 operand value is split into three bf16 parts by round-to-nearest conversions of
// successive remainders,
//       x = h + m + l (+ e),   |m| <= 2^-8 |x|,  |l| <= 2^-16 |x|,  |e| <= 2^-25 |x|  (less than half an fp32 ulp),
// and a product is the six partial products of weight >= 2^-16, each exact in the fp32 accumulator's input:
//       a w ~= al wh + am wm + ah wl + am wh + ah wm + ah wh          (v_mfma_f32_32x32x16_bf16, smallest terms first)
// The three dropped ones (am wl, al wm, al wl) are <= 2^-23 |a w| together: the error per product is about that of ONE
// rounded fp32 multiply (2^-24), and the accumulation is the matrix core's fp32 one, as in gemm_nt.hip -- against fp64 the
// results are closer than the fp32 instructions' (tests/test_gpu_deconv.py::test_gemm_x3_is_as_accurate_as_...: 0.7-0.95x
// their error).  Six bf16 instructions of 32 cycles do the work of sixteen fp32 ones of 32 cycles: a ceiling of
// 2.5 PFLOP/s / 6 = 417 TFLOP/s of fp32 products (tools/split_mfma_probe.hip measures 2.4 PFLOP/s of bare bf16 MFMA).
//
// Around the products the kernel is gemm_nt.hip's: persistent workgroups walking (tile, k range) items as ONE continuous
// sequence of 32-deep k chunks, XCD-aware tile order, data-parallel launch + stream-K tail with fp32 atomics, the epilogues
// (bias, addend, per-group row bias, LeakyReLU, gate, block-shifted BatchNorm partial sums), the result's 16-B stores issued
// inside the next item's first chunk.  The operand path is its own, because the split is vector-ALU work (11 instructions
// per pair of values) that must be done ONCE per value and hidden between the MFMAs:
//   * loaders: every thread reads 16 B (4 consecutive k of one row; a wave instruction = whole 128-B lines) of the chunk
//     AFTER NEXT into registers -- no LDS staging of fp32 data;
//   * conversion: the values loaded one chunk ago are split and written as three bf16 parts, [rows][32 k] each, 64 B per
//     row, into the LDS stage the products will read NEXT (two stages); each quad's registers are reloaded as soon as it is
//     written.  The 16-B column c of row r sits at c ^ ((r >> 2) & 3): conflict-free for the b128 fragment reads' lane
//     groups and for the 8-B writes;
//   * products: a chunk is two 16-deep k steps of 6 TM TN MFMAs; the six partial products run in turn over ALL the wave's
//     32 x 32 blocks (consecutive MFMAs are independent); lane (i, g) of a fragment holds k = 8g .. 8g + 7 of row i: one
//     ds_read_b128 per part, the fragments of the next k step read during the current one;
//   * the conversion tasks, the reloads and the fragment reads sit at fixed places between the MFMAs (a 32 x 32 x 16 bf16
//     MFMA holds the SIMD's vector issue for 8 of its 32 cycles: ~5 single-issue instructions per MFMA hide,
//     MI355X_MICROARCH.md, cycle constants); one barrier per chunk, placed so that the last MFMAs of the chunk cover the
//     first fragment reads of the next; the item's first chunk is peeled off the loop so that the loop body is one code
//     path and the loop-carried registers (values in flight, accumulators) need no copies;
//   * operands given transposed (K x rows: the input- and weight-gradient forms) differ in the loader only: 4 k rows x 4
//     columns per thread, transposed in registers.
// Measured (MI355X, tools/x3_check.py, tools/x3_pmc.sh): 190-200 TFLOP/s on 35840 x 512 x 5120 (gemm_nt.hip: 135) with the
// matrix pipe busy 68 % of the cycles at the 1.77 GHz the chip holds under this load (2.4 GHz nominal: the bf16 peak at
// that clock is 1.84 PFLOP/s = 307 TFLOP/s of fp32 products).
#include <atomic>
#include <type_traits>

#include "gemm_shared.h"

#ifndef X3_FENCE
#define X3_FENCE 2                    // MFMAs per scheduling region of the main loop
#endif
#ifndef X3_DEFAULT_MASK
#define X3_DEFAULT_MASK 0x000         // instance classes on v_mfma_f32_16x16x32_bf16 (nt_switches); PDGN_X3_SHAPE[16_MASK] / pdgn_gemm_set_shape override
#endif
#ifndef X3_FENCE16
#define X3_FENCE16 2                  // the same for the 16x16x32 form
#endif
#ifndef X3_ABLATE
#define X3_ABLATE 0                   // tools/x3_bench.hip (measurement only): 1 no conversion tasks, 2 no operand loads, 4 no stores, 8 no barrier, 16 no fragment reads, 32 no split arithmetic, 64 one LDS write per group, 128 direct (unstaged) result stores, 256 first operand always from the same (cache-resident) block, 512 second operand likewise
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

// NP = 2 (round 5): TWO fp16 parts of the operand scaled by a power of two, x 2^e = h + l (+ err), three partial products
// al wh + ah wl + ah wh on v_mfma_f32_32x32x16_f16 (the kernel's comment at the template parameter NP)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned x2_cvt_pk(float a, float b) {          // v_cvt_pk_f16_f32: round to nearest even
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ f32x16 x2_mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
#ifndef X2_PRODUCTS
#define X2_PRODUCTS 3                 // 4: al wl as well
#endif

typedef float f32x4a __attribute__((ext_vector_type(4)));
// the other bf16 shape, K = 32 in one instruction of 16 cycles: lane (i = l & 15, g = l >> 4) holds k = 8 g .. 8 g + 7 of row i of
// either operand, D[row 4 g + r][column i] in register r
__device__ __forceinline__ f32x4a x3_mfma16(const u32x4 a, const u32x4 b, const f32x4a c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// XOR pattern of the LDS image's 16-B columns for row r.  MS = 32: (r >> 2) & 3, conflict-free for the 32-row fragment reads
// (lanes of a ds_read_b128 group: rows 0-3, 12-15, 20-27 of ONE column).  MS = 16: the groups mix two columns (rows 0-3 and 12-15
// of column g, rows 4-11 of column g ^ 1), which wants f(0), f(3), f(1) ^ 1, f(2) ^ 1 all different: f = (0, 2, 3, 1).  The
// writers (b64: two whole rows per 16 lanes; b128: eight 16-B columns of two rows) are conflict-free under any row pattern.
template <int MS>
__device__ __forceinline__ unsigned x3_sw(unsigned r) {
    const unsigned x = (r >> 2) & 3u;
    return MS == 32 ? x : ((0x78u >> (2u * x)) & 3u);
}

// v + (v of the lane CTRL says), 0 where that lane is outside the row / the row is masked: one v_add_f32 with a DPP operand
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float x3_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}

// Sum over the 32 lanes of each half of the wave, valid in lanes 31 and 63: an inclusive scan inside each row of 16 lanes
// (row_shr 1, 2, 4, 8: lane 15 of a row holds the row's sum), then lane 15 of rows 0 / 2 added into rows 1 / 3 (row_bcast:15).
// Five vector instructions; the ds_bpermute butterfly of __shfl_xor costs an LDS round trip per step.
__device__ __forceinline__ float x3_half_wave_sum(float v) {
    v = x3_dpp_add<0x111, 0xf>(v);
    v = x3_dpp_add<0x112, 0xf>(v);
    v = x3_dpp_add<0x114, 0xf>(v);
    v = x3_dpp_add<0x118, 0xf>(v);
    return x3_dpp_add<0x142, 0xa>(v);
}

// Sum over the 16 lanes of each DPP row, valid in lanes 15, 31, 47, 63.
__device__ __forceinline__ float x3_row_sum(float v) {
    v = x3_dpp_add<0x111, 0xf>(v);
    v = x3_dpp_add<0x112, 0xf>(v);
    v = x3_dpp_add<0x114, 0xf>(v);
    return x3_dpp_add<0x118, 0xf>(v);
}

// TM x TN blocks of 32 x 32 per wave, WM x WN waves, OCC workgroups per CU.
// MS: the matrix instruction.  32: v_mfma_f32_32x32x16_bf16, two 16-deep k steps per chunk, fragments of step s + 1 read during
// step s.  16: v_mfma_f32_16x16x32_bf16 on the SAME wave tile ((2 TM) x (2 TN) blocks of 16 x 16, a chunk is one k step of
// 6 x 4 TM TN instructions of 16 cycles): same LDS image, same number of fragment reads and registers; under the chip's power
// limit this shape holds a higher clock (MI355X_MICROARCH.md, DVFS give-back item 7; cdna_hip_programming.md rule 28).  The six
// partial products of a chunk run in the order  ah wh, ah wm, ah wl, am wh, al wh, am wm  so that every fragment part is
// re-read for the next chunk right after its last use in this one and is back before its first use there -- no second set
// of fragment registers: Wm, Wl, Am during the first product, Al during the next two, and behind the barrier Ah and (when the
// fifth product has issued) Wh of the next chunk.  The k sum of a chunk is taken by one instruction instead of two, and the
// parts are added largest first: results differ from MS = 32 in the last bits (same error bound; tests/test_gpu_deconv.py
// runs both against fp64).
// PW: the second operand arrives PRE-SPLIT -- three bf16 planes [N][ldw] (p.Wp; plane stride p.wplane elements), written once per
// weight by pdgn_split_bf16x3 with the same round-to-nearest remainders the loader computes: its quads are loaded part by part
// (8 B per part and lane) and go to LDS as they are -- none of the 22 vector instructions per quad, a third of the split work
// of a 256 x 128 tile.  Same parts, same products, same order: results are bit-identical to the unsplit operand's.
// NP: parts per operand value.  3: the bf16 form above.  2 (round 5, MS = 32 only): each ROW of each operand (as the kernel sees
// it: a transposed operand's rows are the columns of the matrix in memory) is multiplied by a power of two 2^e_r, e_r = 14 -
// floor(log2 max |x| over that row) (exact; max |x| 2^e_r in [2^14, 2^15): inside fp16's range), and split as x 2^e = h + l with
// h = rn16(x 2^e), l = rn16 of the exact remainder: 22 + 1 significant bits while |x| is within 2^-16 of ITS ROW's largest value,
// an absolute 2^-39 of the row's maximum below (fp16's subnormal spacing).  (Round 5 scaled the whole operand by one power of two:
// a row 2^-26 below the operand's maximum -- a point with a small gradient -- kept 12 bits.  Round 6: per row.)  A product is
// THREE partial products on v_mfma_f32_32x32x16_f16 -- al wh + ah wl + ah wh, each exact in the fp32 accumulator's input -- and
// C[m, n] is multiplied by 2^-e_A[m], then 2^-e_W[n] in front of the epilogue (exact).  Half the matrix-core work, two thirds of the
// LDS image, 7 instead of 11 vector instructions per pair of values in the split; per product |err| <~ 2^-21 |a w| in the worst
// case (al wl dropped + the two representation errors), in a sum far below the fp32 accumulation's own rounding, of which this
// form does half as much: against fp64 its results are the closest of the three forms (tools/x2_check.py).  The row maxima
// arrive as arrays of bit patterns (p.max_a[M] / p.max_w[N]: x2_maxima_kernel, the caller, or the kernel that wrote the operand;
// a pre-split second operand -- two fp16 planes, already scaled row by row -- carries its rows' maxima behind its planes).
// Which launches take this form: x2_pays (host side).
template <int TM, int TN, int WM, int WN, int OCC, bool ATOMIC, bool WT, bool AT, bool EPI = false, bool PW = false, int MS = 32, int NP = 3>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_x3_kernel(const NtArgs p) {
    static_assert(!PW || (!WT && !AT), "pre-split second operand: row-major (N x K) planes only");
    static_assert(MS == 32 || MS == 16, "matrix instruction: 32x32x16 or 16x16x32");
    static_assert(NP == 3 || (NP == 2 && MS == 32), "parts: three bf16, or two fp16 on the 32x32x16 instruction");
    constexpr int NW = WM * WN, NTH = 64 * NW, BM = 32 * TM * WM, BN = 32 * TN * WN;
    // A chunk in LDS: per operand three bf16 parts of [rows][32 k] (64 B per row); the 16-B column c (k = 8c .. 8c + 7) of
    // row r is stored at position c ^ ((r >> 2) & 3): the b128 fragment reads -- serviced in the lane groups {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} (+32) of MI355X_MICROARCH.md's LDS table: the four rows with equal r & 3 of a group differ in
    // (r >> 2) & 3 -- and the b64 writes of the row-major loaders (16 lanes = 2 rows x 64 B) are bank-conflict-free.
    constexpr int PART_A = BM * 64, PART_W = BN * 64, STAGE = NP * (PART_A + PART_W);
    // STG: the result leaves through a per-wave LDS staging block (32 rows x 128 B) so that a store instruction covers whole
    // 128-B lines (8 rows) instead of 32 B of each of 32 rows -- for the one-workgroup-per-CU tiles, whose LDS has the room
    // NP = 2: the item's un-scales 2^-e of its BM rows and BN columns (finish_item reads them from here: held in registers they
    // would be 2 + 32 values per lane over a whole item)
    constexpr int UNT = NP == 2 ? (BM + BN) * 4 : 0;
    constexpr bool STG = OCC == 1 && !ATOMIC && !(X3_ABLATE & 128) && 2 * STAGE + NW * 4096 + UNT <= 160 * 1024;      // (and the CU's 160 KB hold it)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE + (STG ? NW * 4096 : 0) + UNT];
    float *const untab = reinterpret_cast<float *>(smem + 2 * STAGE + (STG ? NW * 4096 : 0));     // [BM row un-scales | BN column un-scales]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = MS == 32 ? lane & 31 : lane & 15, lg = MS == 32 ? lane >> 5 : lane >> 4;   // fragment row, 16-B k column of the lane

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
        if (p.sk_split && (ATOMIC || p.sk_ws)) {                   // split-K: one item per workgroup (atomics into C, or a partial tile into the workspace)
            const int slice = v / p.sk_split;
            c.tile = p.tile_begin + v % p.sk_split;
            c.kc = c.kb = (int)(slice * p.sk_per_wg);
            c.ke = min(KC, c.kb + (int)p.sk_per_wg);
            c.valid = c.j == 0 && c.kb < KC;
            c.j++;
        } else if (ATOMIC || (p.sk_ws && p.sk_seg)) {              // flattened (tile, chunk) order: atomics into C, or partial tiles into the workspace
            c.valid = c.f < c.f1;
            c.j++;                                                 // (1-based index of the item among this workgroup's)
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

    // ---- loaders: global -> registers (one chunk ahead of the conversion, two ahead of the products).  A load is 16 B per
    // lane = 4 values; the 64 lanes of a wave instruction cover whole 128-B lines:
    //   row-major operand: 8 rows x 32 k (lane: row l >> 3, k quad l & 7);
    //   transposed operand (K x rows in memory): 4 loads = k rows 4 kq .. 4 kq + 3 at the same 4 columns, transposed in
    //   registers into 4 rows x 4 k.
    // Either way a thread ends up with "quads": 4 consecutive k of one row = one 8-B write per part.
    // (PW: the second operand's units are OCTETS -- 8 k of one row = one 16-B column of a part's LDS image, 16-B loads)
    constexpr int QA = BM * 8 / NTH, QW = PW ? BN * 4 / NTH : BN * 8 / NTH, NQ = QA + QW;        // quads per thread and chunk
    static_assert(QA * NTH == BM * 8 && QW * NTH == BN * (PW ? 4 : 8), "loader quads must divide over the threads");
    constexpr int CUA = QA % 4 == 0 ? 4 : 2, CUW = QW % 4 == 0 ? 4 : 2;     // columns per transposed unit (16-B or 8-B loads)
    static_assert((!AT || QA % CUA == 0) && (!WT || QW % CUW == 0), "transposed operand: whole units per thread");
    float raw[NQ][4];
    unsigned rawp[PW ? QW : 1][12];                                // PW: octet j of W as (h, m, l) x 4 dwords
    // quad q of the thread: row-major: piece (wave + j NW) of 8 rows; transposed: unit (tid + j NTH) = (k quad, column quad)
    unsigned goffA0, goffW0, woffA0, woffW0;                       // byte offsets of quad 0 (global: inside the tile / chunk; LDS: part 0)
    int kqA, kqW;                                                  // k quad of the thread's row-major quads (one per operand)
    {
        const int r8 = lane >> 3, c16 = lane & 7;
        kqA = kqW = c16;
        if (!AT) {
            const int row = wave * 8 + r8;
            goffA0 = (unsigned)(row * p.lda + 4 * c16) * 4u;
            woffA0 = (unsigned)(row * 64 + (((c16 >> 1) ^ x3_sw<MS>(row)) * 16) + (c16 & 1) * 8);
        } else {
            const int cq = tid % (BM / CUA), kq = tid / (BM / CUA);          // units per chunk: 8 k quads x BM / CUA column groups
            goffA0 = (unsigned)((4 * kq) * p.lda + CUA * cq) * 4u;
            woffA0 = 0;                                            // (computed per quad in conv_write)
        }
        if (PW) {
            const int row = tid >> 2, col = tid & 3;               // octet j: row + j NTH / 4 (a multiple of 16: swizzle unchanged)
            kqW = 2 * col;
            goffW0 = (unsigned)(row * p.ldw + 8 * col) * 2u;
            woffW0 = (unsigned)(NP * PART_A + row * 64 + ((col ^ x3_sw<MS>(row)) * 16));
        } else if (!WT) {
            const int row = wave * 8 + r8;
            goffW0 = (unsigned)(row * p.ldw + 4 * c16) * 4u;
            woffW0 = (unsigned)(NP * PART_A + row * 64 + (((c16 >> 1) ^ x3_sw<MS>(row)) * 16) + (c16 & 1) * 8);
        } else {
            const int cq = tid % (BN / CUW), kq = tid / (BN / CUW);
            goffW0 = (unsigned)((4 * kq) * p.ldw + CUW * cq) * 4u;
            woffW0 = 0;
        }
    }
    // transposed: the thread's units are NTH apart in unit index = NTH / (B / CU) k quads further (B / CU column groups per k quad)
    constexpr int KQ_STEP_A = NTH / (BM / CUA > NTH ? NTH : BM / CUA), KQ_STEP_W = NTH / (BN / CUW > NTH ? NTH : BN / CUW);
    static_assert(!AT || (BM / CUA <= NTH && NTH % (BM / CUA) == 0), "A^T units");
    static_assert(!WT || (BN / CUW <= NTH && NTH % (BN / CUW) == 0), "W^T units");

    __amdgpu_buffer_rsrc_t rsA, rsW;                               // descriptors of the load cursor's chunk
    long long ld_m0 = 0;
    int ld_n0 = 0, ld_mrows = 0, ld_nrows = 0;
    // NP = 2: the power-of-two scale of each ROW the thread converts (row-major operand: one per quad; transposed: the CU rows of
    // its units; a pre-split second operand arrives scaled).  scA / scW belong to the item whose chunk sits in the raw registers;
    // the maxima of the load cursor's NEXT item are fetched (nxA / nxW) when that item's first chunk is about to be loaded and
    // become the scales once the last chunk of the item before has been converted (the chunk loop's hooks, below).
    constexpr int NSA = NP == 2 ? (AT ? CUA : QA) : 1, NSW = (NP == 2 && !PW) ? (WT ? CUW : QW) : 1;
    float scA[NSA], scW[NSW];
    unsigned nxA[NSA], nxW[NSW];
    auto make_srds = [&](int tile) {
        int tm, tn;
        decode(tile, tm, tn);
        ld_m0 = (long long)tm * BM;
        ld_n0 = tn * BN;
        ld_mrows = (int)min((long long)BM, p.M - ld_m0);
        ld_nrows = min(BN, p.N - ld_n0);
    };
    make_srds(ld.tile);
    // the row maxima of the load cursor's tile (rows past the operand read as 0: scale 2^126 of values that are zero)
    auto fetch_maxima = [&]() {
        if (NP != 2) return;
        int lane_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_));
        const int tid_ = wave * 64 + lane_;
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void *)(p.max_a + ld_m0), 0, ld_mrows * 4, 0x00020000);
        if (!AT) {
#pragma unroll
            for (int j = 0; j < NSA; ++j) nxA[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rA, (unsigned)(wave * 8 + (lane_ >> 3) + j * NW * 8) * 4u, 0, 0);
        } else if (CUA == 4) {
            const u32x4 x = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, (unsigned)(tid_ % (BM / CUA)) * 16u, 0, 0));
#pragma unroll
            for (int j = 0; j < NSA; ++j) nxA[j] = x[j];
        } else {
            typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
            const u32x2_ x = __builtin_bit_cast(u32x2_, __builtin_amdgcn_raw_buffer_load_b64(rA, (unsigned)(tid_ % (BM / CUA)) * 8u, 0, 0));
#pragma unroll
            for (int j = 0; j < NSA; ++j) nxA[j] = x[j];
        }
        if (PW) return;
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void *)(p.max_w + ld_n0), 0, ld_nrows * 4, 0x00020000);
        if (!WT) {
#pragma unroll
            for (int j = 0; j < NSW; ++j) nxW[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rW, (unsigned)(wave * 8 + (lane_ >> 3) + j * NW * 8) * 4u, 0, 0);
        } else if (CUW == 4) {
            const u32x4 x = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, (unsigned)(tid_ % (BN / CUW)) * 16u, 0, 0));
#pragma unroll
            for (int j = 0; j < NSW; ++j) nxW[j] = x[j];
        } else {
            typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
            const u32x2_ x = __builtin_bit_cast(u32x2_, __builtin_amdgcn_raw_buffer_load_b64(rW, (unsigned)(tid_ % (BN / CUW)) * 8u, 0, 0));
#pragma unroll
            for (int j = 0; j < NSW; ++j) nxW[j] = x[j];
        }
    };
    auto adopt_scales = [&]() {                                    // the fetched maxima become the conversion's scales
        if (NP != 2) return;
#pragma unroll
        for (int j = 0; j < NSA; ++j) scA[j] = x2_scale(nxA[j]);
        if (PW) return;
#pragma unroll
        for (int j = 0; j < NSW; ++j) scW[j] = x2_scale(nxW[j]);
    };
    int ld_k0 = 0;
    bool kokA = true, kokW = true;
    auto issue_begin = [&]() {                                     // the load cursor's chunk; past the end of the sequence: empty descriptors
        ld_k0 = ld.kc * NT_BK;
        kokA = ld_k0 + 4 * kqA < p.K;                              // K % 4 == 0: a 16-B column is all in or all out
        kokW = ld_k0 + 4 * kqW < p.K;
        const int krows = min(NT_BK, p.K - ld_k0);
        const bool ok = ld.valid;
        if (!AT && (X3_ABLATE & 256)) rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.A, 0, ok ? 0x7fffffff : 0, 0x00020000);   // (measurement: every chunk of every tile reads the SAME 256 x 32 block: cache hits)
        else if (!AT) rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(p.A + ld_m0 * p.lda + ld_k0), 0,
                                                         ok ? (int)(((long long)(ld_mrows - 1) * p.lda + (p.K - ld_k0)) * 4) : 0, 0x00020000);
        else rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(p.A + (long long)ld_k0 * p.lda + ld_m0), 0,
                                                     ok ? (int)(((long long)(krows - 1) * p.lda + ld_mrows) * 4) : 0, 0x00020000);
        // PW: ONE descriptor from the tile's start in plane h to the end of plane l; the plane offset of a load travels in the
        // instruction's SCALAR offset, which the hardware's range check does not see (raw buffers: voffset + immediate only):
        // rows past the operand's last one are masked per lane instead (load_quad: rowokW), so that no plane is read past its
        // N rows; rows inside the operand but past the TILE cannot occur (a tile's rows are min(BN, N - n0)); the K tail is masked
        // by kokW as always
        if (PW) rsW = __builtin_amdgcn_make_buffer_rsrc((void *)(p.Wp + (long long)ld_n0 * p.ldw + ld_k0), 0,
                                                        ok ? (int)(((NP - 1) * p.wplane + (long long)(p.N - 1 - ld_n0) * p.ldw + (p.K - ld_k0)) * 2) : 0, 0x00020000);
        else if (!WT && (X3_ABLATE & 512)) rsW = __builtin_amdgcn_make_buffer_rsrc((void *)p.W, 0, ok ? 0x7fffffff : 0, 0x00020000);
        else if (!WT) rsW = __builtin_amdgcn_make_buffer_rsrc((void *)(p.W + (long long)ld_n0 * p.ldw + ld_k0), 0,
                                                         ok ? (int)(((long long)(ld_nrows - 1) * p.ldw + (p.K - ld_k0)) * 4) : 0, 0x00020000);
        else rsW = __builtin_amdgcn_make_buffer_rsrc((void *)(p.W + (long long)ld_k0 * p.ldw + ld_n0), 0,
                                                     ok ? (int)(((long long)(krows - 1) * p.ldw + ld_nrows) * 4) : 0, 0x00020000);
    };
    // the load(s) that fill quad q (transposed: the 4 loads of its unit, issued with the unit's first quad)
    auto load_quad = [&](int q) {
        const bool isA = q < QA;
        const bool tr = isA ? AT : WT;
        const int j = isA ? q : q - QA;
        const __amdgpu_buffer_rsrc_t rs = isA ? rsA : rsW;
        const int ld_ = isA ? p.lda : p.ldw;
        const unsigned g0 = isA ? goffA0 : goffW0;
        if (PW && !isA) {
            // parts h, m, l of the quad: the same tile-relative offset in each plane, the plane offset in the scalar offset
            const unsigned off = g0 + (unsigned)(j * (NTH / 4) * ld_) * 2u;
            const bool rowok = (int)(tid >> 2) + j * (NTH / 4) < ld_nrows;          // (the scalar plane offset escapes the range check)
#pragma unroll
            for (int part = 0; part < NP; ++part) {
                const u32x4 x = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (kokW && rowok) ? off : NT_OOB, (int)(part * p.wplane * 2), 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) rawp[j][4 * part + e] = x[e];
            }
        } else if (!tr) {
            const unsigned off = g0 + (unsigned)(j * NW * 8 * ld_) * 4u;
            const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (isA ? kokA : kokW) ? off : NT_OOB, 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) raw[q][e] = x[e];
        } else if ((isA ? CUA : CUW) == 4) {
            if ((j & 3) == 0) {
                const unsigned off = g0 + (unsigned)((j >> 2) * (isA ? KQ_STEP_A : KQ_STEP_W) * 4 * ld_) * 4u;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)(kk * ld_) * 4u, 0, 0));
#pragma unroll
                    for (int c = 0; c < 4; ++c) raw[q + c][kk] = x[c];
                }
            }
        } else if ((j & 1) == 0) {
            const unsigned off = g0 + (unsigned)((j >> 1) * (isA ? KQ_STEP_A : KQ_STEP_W) * 4 * ld_) * 4u;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x2 x = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off + (unsigned)(kk * ld_) * 4u, 0, 0));
                raw[q][kk] = x[0];
                raw[q + 1][kk] = x[1];
            }
        }
    };
    auto advance_load = [&]() {
        if (!ld.valid) return;
        ld.kc++;
        if (ld.kc == ld.ke) {
            next_item(ld);
            if (ld.valid) make_srds(ld.tile);
        }
    };

    // ---- conversion: the 4 values of quad q -> three 8-B writes into stage `st`
    unsigned cvh[2], cvm[2], cvl[2];
    auto conv_pair = [&](int q, int e) {
        if (PW && q >= QA) return;                                 // already parts
        const float a = raw[q][2 * e], b = raw[q][2 * e + 1];
        if (X3_ABLATE & 32) {                                      // (measurement: no arithmetic)
            cvh[e] = __float_as_uint(a);
            cvm[e] = __float_as_uint(b);
            cvl[e] = cvh[e] ^ cvm[e];
            return;
        }
        if (NP == 2) {
            // scaled into fp16's range (exact: a power of two), h = rn16, l = rn16 of the exact remainder
            const float sc = q < QA ? scA[AT ? q % CUA : q] : scW[PW ? 0 : (WT ? (q - QA) % CUW : q - QA)];      // the quad's row
            const float a2 = a * sc, b2 = b * sc;
            const unsigned h = x2_cvt_pk(a2, b2);
            const f16x2 hh = __builtin_bit_cast(f16x2, h);
            cvh[e] = h;
            cvl[e] = x2_cvt_pk(a2 - (float)hh[0], b2 - (float)hh[1]);
            return;
        }
        const unsigned h = x3_cvt_pk(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);      // exact
        const unsigned m = x3_cvt_pk(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);    // exact
        cvh[e] = h;
        cvm[e] = m;
        cvl[e] = x3_cvt_pk(sa, sb);
    };
    auto conv_write = [&](int st, int q) {
        const bool isA = q < QA;
        const bool tr = isA ? AT : WT;
        const int j = isA ? q : q - QA;
        const int ps = isA ? PART_A : PART_W;
        unsigned off = isA ? woffA0 : woffW0;
        if (PW && !isA) {
            off += (unsigned)(j * (NTH / 4) * 64);                 // octets: NTH / 4 rows apart
        } else if (!tr) {
            off += (unsigned)(j * NW * 8 * 64);                    // 8 rows per piece, pieces NW apart (32 rows): (row >> 2) & 3 unchanged
        } else {
            // unit j / CU: k quad kq0 + (j / CU) KQ_STEP: 16-B column (kq >> 1) ^ swizzle, half kq & 1; row CU cq + j % CU
            const int CU = isA ? CUA : CUW;
            const int d = (j / CU) * (isA ? KQ_STEP_A : KQ_STEP_W);
            const int kq0 = isA ? (int)(tid / (BM / CUA)) : (int)(tid / (BN / CUW));
            const int cq = isA ? (int)(tid % (BM / CUA)) : (int)(tid % (BN / CUW));
            const int row = CU * cq + (j % CU), kq = kq0 + d;
            off = (unsigned)((isA ? 0 : NP * PART_A) + row * 64 + (((kq >> 1) ^ x3_sw<MS>(row)) * 16) + (kq & 1) * 8);
        }
        unsigned char *dst = smem + st * STAGE + off;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        if (PW && !isA) {
#pragma unroll
            for (int part = 0; part < NP; ++part)
                *reinterpret_cast<u32x4 *>(dst + part * ps) = (u32x4){rawp[j][4 * part], rawp[j][4 * part + 1], rawp[j][4 * part + 2], rawp[j][4 * part + 3]};
            return;
        }
        if (X3_ABLATE & 64) {                                      // (measurement: one write instead of three)
            *reinterpret_cast<u32x2 *>(dst) = (u32x2){cvh[0] ^ cvm[0] ^ cvl[0], cvh[1] ^ cvm[1] ^ cvl[1]};
            return;
        }
        *reinterpret_cast<u32x2 *>(dst) = (u32x2){cvh[0], cvh[1]};
        if (NP == 2) {
            *reinterpret_cast<u32x2 *>(dst + ps) = (u32x2){cvl[0], cvl[1]};
            return;
        }
        *reinterpret_cast<u32x2 *>(dst + ps) = (u32x2){cvm[0], cvm[1]};
        *reinterpret_cast<u32x2 *>(dst + 2 * ps) = (u32x2){cvl[0], cvl[1]};
    };

    // ---- fragments.  MS = 32: lane (li, lg) of k step s holds k = 16 s + 8 lg .. + 7 of row li of a 32-row block: column
    // 2 s + lg.  MS = 16: lane (li, lg) holds k = 8 lg .. + 7 of row li of a 16-row block: column lg, the whole chunk.
    constexpr int FR = MS == 32 ? 32 : 16;                         // rows of a fragment / of an output block
    constexpr int FA = 32 * TM / FR, FW = 32 * TN / FR, NPAR = MS == 32 ? 2 : 1;
    const unsigned a_rd = (unsigned)((wm * 32 * TM + li) * 64 + ((lg ^ x3_sw<MS>(li)) * 16));
    const unsigned w_rd = (unsigned)(NP * PART_A + (wn * 32 * TN + li) * 64 + ((lg ^ x3_sw<MS>(li)) * 16));
    X3Parts fa[NPAR][FA], fw[NPAR][FW];                            // MS = 32: [parity of the k step]
    // fragment read r of k step s (A blocks first, 3 parts each) from stage st into parity `par`
    auto read_frag = [&](int st, int s, int par, int r) {
        const int f = r / NP, part = NP == 2 ? 2 * (r % NP) : r % NP;      // (NP = 2: parts h and l)
        if (f < FA) {
            const u32x4 x = *reinterpret_cast<const u32x4 *>(smem + st * STAGE + ((a_rd ^ (unsigned)(32 * s)) + f * FR * 64 + (NP == 2 ? part / 2 : part) * PART_A));
            if (part == 0) fa[par][f].h = x;
            else if (part == 1) fa[par][f].m = x;
            else fa[par][f].l = x;
        } else {
            const int b = f - FA;
            const u32x4 x = *reinterpret_cast<const u32x4 *>(smem + st * STAGE + ((w_rd ^ (unsigned)(32 * s)) + b * FR * 64 + (NP == 2 ? part / 2 : part) * PART_W));
            if (part == 0) fw[par][b].h = x;
            else if (part == 1) fw[par][b].m = x;
            else fw[par][b].l = x;
        }
    };
    // MS = 16: one part (0 h, 1 m, 2 l) of fragment f of operand A / W
    auto read_part = [&](int st, bool isA, int part, int f) {
        read_frag(st, 0, 0, 3 * (isA ? f : FA + f) + part);
    };

    // accumulators: MS = 32: acc[a][b] = 32 x 32 block (a, b), 16 registers; MS = 16: the same registers as four 16 x 16 blocks
    // (2 a + ar, 2 b + bc) -> acc[a][b][4 (2 ar + bc) .. + 3]
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- output of a finished item.  A lane holds 4 consecutive output columns of one output row per "sub-block":
    //   MS = 32: sub-block bb = 4 b + q of block row a: row 32 a + li, columns 32 b + 8 q + 4 lg + (0 .. 3);
    //   MS = 16: sub-block bb (= 16-column block) of block row a (16 rows): row 16 a + li, columns 16 bb + 4 lg + (0 .. 3).
    // The 16-B stores of item t are issued from inside the first k step of item t + 1 (before the MFMA that restarts the block
    // from zero), in the shadow of running MFMAs.
    constexpr int NBB = MS == 32 ? 4 * TN : 2 * TN, NA = FA;      // (mloc0 / nloc0, the lane's first row / column: finish_item derives them)
    auto coloff = [](int bb) { return MS == 32 ? 32 * (bb >> 2) + 8 * (bb & 3) : 16 * bb; };
    auto get4 = [&](int a, int bb) {
        const int A_ = MS == 32 ? a : a >> 1, b = MS == 32 ? bb >> 2 : bb >> 1, q = MS == 32 ? bb & 3 : 2 * (a & 1) + (bb & 1);
        return (f32x4){acc[A_][b][4 * q], acc[A_][b][4 * q + 1], acc[A_][b][4 * q + 2], acc[A_][b][4 * q + 3]};
    };
    auto set4 = [&](int a, int bb, const f32x4 x) {
        const int A_ = MS == 32 ? a : a >> 1, b = MS == 32 ? bb >> 2 : bb >> 1, q = MS == 32 ? bb & 3 : 2 * (a & 1) + (bb & 1);
        acc[A_][b][4 * q] = x[0]; acc[A_][b][4 * q + 1] = x[1]; acc[A_][b][4 * q + 2] = x[2]; acc[A_][b][4 * q + 3] = x[3];
    };
    // p.sk_ws (stream-K tail without atomics, round 5): the workgroup's partial tile goes to slot v of a workspace of whole BM x BN
    // tiles (pitch BN) through the ordinary store path; x3_sk_reduce_kernel adds the slices of a tile and writes C
    const int ldc_eff = (!ATOMIC && p.sk_ws) ? BN : p.ldc;
    __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)p.C, 0, 0, 0x00020000);   // nothing pending: all out of range
    unsigned st_off[NBB];                                          // byte offset of (row mloc0, sub-block bb), or out of range
#pragma unroll
    for (int bb = 0; bb < NBB; ++bb) st_off[bb] = NT_OOB;
    // staged form: lane l stores row 8 j + (l >> 3), columns 4 (l & 7) .. + 3 of a 32 x 32 block (j = 0 .. 3)
    unsigned stg_off[TN];                                          // byte offset of (row wm 32 TM + (l >> 3), column block b), or out of range
#pragma unroll
    for (int b = 0; b < TN; ++b) stg_off[b] = NT_OOB;
    unsigned char *const stg = smem + 2 * STAGE + wave * 4096;
    // writer: MS = 32: row li, 16-B column (2 q + lg) ^ (li & 7); MS = 16: row 16 ar + li, 16-B column (4 bc + lg) ^ (li & 7);
    // reader: lane l takes row 8 j + (l >> 3), 16-B column (l & 7) ^ (l >> 3)
    u32x4 pend[4];                                                 // a block on its way out: read back from the staging block
    // (a, b): a 32 x 32 block of the wave tile, for both instruction shapes
    auto stage_block = [&](int a, int b) {                         // accumulators -> LDS -> pend (LDS operations are in order)
        // (lane geometry from the hardware lane id, as in finish_item: nothing of it is carried through the chunk loop)
        int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane));
        const int li = MS == 32 ? lane & 31 : lane & 15, lg = MS == 32 ? lane >> 5 : lane >> 4;
        const unsigned stg_wr = (unsigned)(li * 128), stg_sw = (unsigned)(li & 7);
        const unsigned stg_rd = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) * 16));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MS == 32)
                *reinterpret_cast<u32x4 *>(stg + stg_wr + (((unsigned)(2 * q + lg) ^ stg_sw) * 16)) = __builtin_bit_cast(u32x4, get4(a, 4 * b + q));
            else
                *reinterpret_cast<u32x4 *>(stg + stg_wr + (q >> 1) * 2048 + (((unsigned)(4 * (q & 1) + lg) ^ stg_sw) * 16)) =
                    __builtin_bit_cast(u32x4, get4(2 * a + (q >> 1), 2 * b + (q & 1)));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) pend[j] = *reinterpret_cast<const u32x4 *>(stg + stg_rd + j * 1024);
    };
    auto flush_block = [&](int a, int b) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!(X3_ABLATE & 4))
                __builtin_amdgcn_raw_buffer_store_b128(pend[j], rsC, stg_off[b] + (unsigned)((a * 32 + 8 * j) * ldc_eff) * 4u, 0, 0);
    };
    auto store_block = [&](int a, int b) {
        if (STG) {
            stage_block(a, b);
            flush_block(a, b);
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ra = MS == 32 ? a : 2 * a + (q >> 1), bb = MS == 32 ? 4 * b + q : 2 * b + (q & 1);
            const unsigned off = st_off[bb] + (unsigned)(ra * FR * ldc_eff) * 4u;
            if (!(X3_ABLATE & 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, get4(ra, bb)), rsC, off, 0, 0);
        }
    };
    auto finish_item = [&]() {
        // The lane geometry the epilogue needs is derived HERE from the hardware lane id (v_mbcnt_*: no live-in register, opaque
        // to common-subexpression elimination): held from the prologue these values stayed in registers -- on the 256 x 128 tiles
        // in scratch -- across the whole item (tools/spill_table.py).
        int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane));
        const int li = MS == 32 ? lane & 31 : lane & 15, lg = MS == 32 ? lane >> 5 : lane >> 4;
        const int mloc0 = wm * 32 * TM + li, nloc0 = wn * 32 * TN + 4 * lg;
        int tm, tn;
        decode(cp.tile, tm, tn);
        const long long m0 = (long long)tm * BM;
        const int n0 = tn * BN;
        const bool to_ws = !ATOMIC && p.sk_ws != nullptr;
        if (NP == 2) {
            // out of the scaled domain: row m by 2^-e_A[m], column n by 2^-e_W[n] (exact; one after the other: the product of the two
            // factors may be outside fp32's range, the result is not).  The factors come from the item's table in LDS
            // (store_unscales, written behind the barrier of the item's first chunk).
            if (!ATOMIC) {
                float ua[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a) ua[a] = untab[mloc0 + 32 * a];
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb) {
                    const f32x4 uw = *reinterpret_cast<const f32x4 *>(untab + BM + nloc0 + coloff(bb));
#pragma unroll
                    for (int a = 0; a < TM; ++a) set4(a, bb, (get4(a, bb) * ua[a]) * uw);
                }
            } else {                                               // operands swapped: D row 8 q + 4 lg + r = activation row, D column li = weight row
                const int mla = wm * 32 * TM + 4 * lg, nla = wn * 32 * TN + li;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const float uw = untab[BM + nla + 32 * b];
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 ua4 = *reinterpret_cast<const f32x4 *>(untab + mla + 32 * a + 8 * q);
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[a][b][4 * q + r] = (acc[a][b][4 * q + r] * ua4[r]) * uw;
                        }
                }
            }
        }
        const long long mrows = to_ws ? BM : min((long long)BM, p.M - m0);      // (a partial tile is stored whole: rows / columns past
        const int ncols = to_ws ? BN : min(BN, p.N - n0);                       //  the matrix hold zeros and are not read back)
        if (!ATOMIC) {
            if (to_ws) rsC = __builtin_amdgcn_make_buffer_rsrc((void *)(p.sk_ws + (size_t)(p.sk_seg ? v * p.sk_seg + (cp.j - 1) : v) * BM * BN), 0,
                                                               BM * BN * 4, 0x00020000);
            else rsC = __builtin_amdgcn_make_buffer_rsrc((void *)(p.C + m0 * p.ldc + n0), 0, (int)(mrows * p.ldc * 4), 0x00020000);
#pragma unroll
            for (int bb = 0; bb < NBB; ++bb)
                st_off[bb] = (nloc0 + coloff(bb) < ncols && !(p.dbg & 1)) ? (unsigned)(mloc0 * ldc_eff + nloc0 + coloff(bb)) * 4u : NT_OOB;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int nl = wn * 32 * TN + 32 * b + 4 * (lane & 7);
                stg_off[b] = (nl < ncols && !(p.dbg & 1)) ? (unsigned)((wm * 32 * TM + (lane >> 3)) * ldc_eff + nl) * 4u : NT_OOB;
            }
            if (p.addend) {                                        // out of range reads 0
                __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)(p.addend + m0 * p.ldadd + n0), 0,
                                                                               (int)(mrows * p.ldadd * 4), 0x00020000);
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb)
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        const int ml = mloc0 + FR * a, nl = nloc0 + coloff(bb);
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
                    for (int a = 0; a < NA; ++a) set4(a, bb, get4(a, bb) + bz);
                }
            }
            if (EPI && p.row_bias) {                               // a bias per group of rows: the heads' per-sample term
                // (bounded by the table's own extent -- groups x ld_rb floats, < 2^30 bytes by the entry point's check -- so that the
                // masked lanes' NT_OOB offset is out of range: an unbounded descriptor made them read row_bias + 1 GiB)
                __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)p.row_bias, 0, p.rb_bytes, 0x00020000);
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const unsigned row = (unsigned)(m0 + mloc0 + FR * a);
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
                    for (int a = 0; a < NA; ++a) {
                        const int ml = mloc0 + FR * a, nl = nloc0 + coloff(bb);
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
                    for (int r = 0; r < 4; ++r) {                  // row li = 0 of block a = 0: the first lane of every lg
                        if (MS == 32) {
                            const float p0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x0[r]), 0));
                            const float p1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x0[r]), 32));
                            pv[r] = lg ? p1 : p0;
                        } else {
                            const float p0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x0[r]), 0));
                            const float p1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x0[r]), 16));
                            const float p2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x0[r]), 32));
                            const float p3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x0[r]), 48));
                            pv[r] = (lg & 2) ? ((lg & 1) ? p3 : p2) : ((lg & 1) ? p1 : p0);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < NA; ++a)
                        if (mloc0 + FR * a < mrows) {
                            const f32x4 x = get4(a, bb);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float d = x[r] - pv[r];
                                cs[r] += d;
                                cq[r] = __fmaf_rn(d, d, cq[r]);
                            }
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                  // DPP adds: the sums end up in the last lane of every lg
                        cs[r] = MS == 32 ? x3_half_wave_sum(cs[r]) : x3_row_sum(cs[r]);
                        cq[r] = MS == 32 ? x3_half_wave_sum(cq[r]) : x3_row_sum(cq[r]);
                    }
                    const int nl = nloc0 + coloff(bb);
                    if (li == FR - 1 && nl < ncols) {
                        *reinterpret_cast<float4 *>(P + n0 + nl) = make_float4(cs[0], cs[1], cs[2], cs[3]);
                        *reinterpret_cast<float4 *>(P + p.N + n0 + nl) = make_float4(cq[0], cq[1], cq[2], cq[3]);
                        *reinterpret_cast<float4 *>(P + 2 * p.N + n0 + nl) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                    }
                }
            }
        } else {
            // partial tile (operands swapped): MS = 32: D row (8q + 4 lg + r) = activation row, D column li = weight row: 128-B
            // atomic segments; MS = 16: D row (4 lg + r), D column li, per 16 x 16 block: 64-B segments.  The holder of the
            // tile's first chunk adds the bias / addend.
            const bool head = cp.kb == 0;
            const int mla = wm * 32 * TM + 4 * lg, nla = wn * 32 * TN + li;
#pragma unroll
            for (int b = 0; b < FW; ++b) {
                const int nl = nla + FR * b;
                const bool nok = nl < ncols;
                const float bz = (head && p.bias && nok) ? p.bias[n0 + nl] : 0.f;
#pragma unroll
                for (int a = 0; a < FA; ++a)
#pragma unroll
                    for (int j = 0; j < (MS == 32 ? 16 : 4); ++j) {
                        const int ml = mla + FR * a + (MS == 32 ? 8 * (j >> 2) + (j & 3) : j);
                        if (nok && ml < mrows) {
                            float o = (MS == 32 ? acc[a][b][j] : acc[a >> 1][b >> 1][4 * (2 * (a & 1) + (b & 1)) + j]) + bz;
                            if (head && p.addend) o += p.addend[(m0 + ml) * p.ldadd + n0 + nl];
                            atomicAdd(p.C + (m0 + ml) * p.ldc + n0 + nl, o);
                        }
                    }
            }
        }
    };

    // ---- one chunk: two k steps of 6 TM TN MFMAs, each step: the six partial products in turn (smallest first), every one
    // over all the wave's blocks (consecutive MFMAs are independent).  Between the MFMAs, at fixed places and fenced against
    // the compiler's scheduler:
    //   * the conversion of the NEXT chunk (in the raw registers) into the other stage, quad by quad, each quad's registers
    //     reloaded with the chunk after next as soon as it is written;
    //   * the fragment reads of the next k step;
    //   * BAR MFMAs before the end: the barrier (the next chunk's parts complete and visible, this stage read to the end), after
    //     which the remaining MFMAs cover the first fragment reads of the next chunk.
    constexpr int NPROD = NP == 3 ? 6 : X2_PRODUCTS;               // partial products per fp32 product
    constexpr int SM = NPROD * TM * TN, NM = 2 * SM, NFR = NP * (TM + TN);   // MFMAs per k step / chunk, fragment reads per step
    constexpr int BAR = 2 * TM * TN > NFR + 2 ? 2 * TM * TN : NFR + 2;       // MFMAs after the barrier
    constexpr int CT_PER_Q = 4, NCT = NQ * CT_PER_Q;               // conversion tasks: per quad 2 pairs, the writes, the reload
    constexpr int C_LO = 1, C_HI = NM - BAR - 1;                   // MFMA gaps that take them
    static_assert(MS == 16 || (BAR < SM && NFR + 2 <= SM), "k step too short for its fragment reads");
    auto conv_task = [&](int st, int k) {
        const int q = k / CT_PER_Q, sub = k % CT_PER_Q;
        if (X3_ABLATE & 1) return;
        if (sub < 2) conv_pair(q, sub);
        else if (sub == 2) conv_write(st, q);
        else {
            const bool tr = q < QA ? AT : WT;
            if (X3_ABLATE & 2) return;
            const int CU = q < QA ? CUA : CUW, j = q < QA ? q : q - QA;
            if (!tr) load_quad(q);
            else if (j % CU == CU - 1) load_quad(q - (CU - 1));                  // after the unit's last quad
        }
    };
    auto chunk = [&](auto first_c, int st) {                       // FIRST: the item's first chunk restarts the accumulators
        constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < NPROD; ++t)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const int is = (t * TM + a) * TN + b, idx = s * SM + is;
                        if (idx == NM - BAR) {
                            // the barrier: my conversion writes and fragment reads are done; then everyone's
                            __builtin_amdgcn_sched_barrier(0);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            if (!(X3_ABLATE & 8)) __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        f32x16 c = acc[a][b];
                        if (FIRST && s == 0 && t == 0) {           // the pending stores of the block, then its restart from zero
                            if (STG) {
                                // staged: the previous block's lines go out (read back one MFMA ago), then this block enters the
                                // staging block; the last one is flushed after the first MFMA of the second product
                                if (a + b > 0) flush_block(b ? a : a - 1, b ? b - 1 : TN - 1);
                                stage_block(a, b);
                            } else if (!ATOMIC) {
                                store_block(a, b);
                            }
#pragma unroll
                            for (int r = 0; r < 16; ++r) c[r] = 0.f;
                        }
                        if (FIRST && STG && s == 0 && t == 1 && a == 0 && b == 0) flush_block(TM - 1, TN - 1);
                        const X3Parts &A = fa[s][a], &W = fw[s][b];
                        if (NP == 2) {                             // (al wl,) al wh, ah wl, ah wh
                            const int u = t + 4 - NPROD;
                            const u32x4 ap = u <= 1 ? A.l : A.h, wp = (u == 0 || u == 2) ? W.l : W.h;
                            acc[a][b] = ATOMIC ? x2_mfma(ap, wp, c) : x2_mfma(wp, ap, c);
                        } else {
                        const u32x4 ap = t == 0 ? A.l : (t == 1 || t == 3) ? A.m : A.h;      // al wh, am wm, ah wl, am wh, ah wm, ah wh
                        const u32x4 wp = t == 2 ? W.l : (t == 1 || t == 4) ? W.m : W.h;
                        acc[a][b] = ATOMIC ? x3_mfma(ap, wp, c) : x3_mfma(wp, ap, c);
                        }
                        // fillers after MFMA idx; the scheduler's regions are X3_FENCE MFMAs long
                        if (idx % X3_FENCE == 0) __builtin_amdgcn_sched_barrier(0);
                        if (!(X3_ABLATE & 16)) {
                            // k step 0: the fragments of step 1, from its second MFMA on; step 1: of the next chunk's step 0 (other
                            // stage), right after the barrier
                            if (s == 0 && is >= 1 && is < 1 + NFR) read_frag(st, 1, 1, is - 1);
                            if (s == 1 && idx >= NM - BAR && idx < NM - BAR + NFR) read_frag(st ^ 1, 0, 0, idx - (NM - BAR));
                        }
                        if (idx >= C_LO && idx < C_HI) {
                            const int i0 = idx - C_LO, span = C_HI - C_LO;
                            const int k0 = (i0 * NCT) / span, k1 = ((i0 + 1) * NCT) / span;
#pragma unroll
                            for (int k = k0; k < k1; ++k) conv_task(st ^ 1, k);
                        }
                    }
    };

    // ---- the same chunk on v_mfma_f32_16x16x32_bf16: ONE k step of 6 FA FW instructions; product t over all the wave's 16 x 16
    // blocks, walked 32 x 32 block by 32 x 32 block (four instructions each: the unit of the result's staging).  Parts:
    //   t:  0 ah wh | 1 ah wm | 2 ah wl | 3 am wh | 4 al wh | 5 am wm        last use: Ah, Wl: t2; Wh, Al: t4; Am, Wm: t5
    // so the parts of THIS chunk that the previous one used to its end (Wm, Am) or that are needed late (Wl, Al) are read from the
    // first product on, one every SP instructions, in the order of their first use (Wm t1, Wl t2, Am t3, Al t4); behind the barrier
    // -- BAR16 instructions before the end, after Ah's last use -- Ah of the NEXT chunk, and from the first instruction of the last
    // product on (Wh's last use is t4) its Wh.  One set of fragment registers.
    constexpr int PM = FA * FW, NM16 = 6 * PM;
    constexpr int SP = PM >= 32 ? 4 : (PM >= 16 ? 2 : 1);
    constexpr int BAR16 = 2 * FA + 4 > NM16 / 5 ? 2 * FA + 4 : NM16 / 5;
    constexpr int C_HI16 = NM16 - BAR16 - 1;
    constexpr int L1 = 2 * FW + 2 * FA;
    static_assert(MS == 32 || (1 + SP * (FW - 1) + 6 <= PM && 1 + SP * (2 * FW - 1) + 6 <= 2 * PM && 1 + SP * (2 * FW + FA - 1) + 6 <= 3 * PM &&
                               1 + SP * (L1 - 1) + 6 <= 4 * PM && 1 + SP * (L1 - 1) < NM16 - BAR16),
                  "16x16x32: a part would not be back before its first use");
    static_assert(MS == 32 || (NM16 - BAR16 >= 3 * PM && NM16 - BAR16 + 1 + 2 * (FA - 1) < NM16 && 5 * PM + 2 * (FW - 1) < NM16),
                  "16x16x32: the next chunk's first parts do not fit behind the barrier");
    auto chunk16 = [&](auto first_c, int st) {
        constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int sb = 0; sb < PM / 4; ++sb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = t * PM + 4 * sb + q;
            const int a32 = sb / TN, b32 = sb % TN, a16 = 2 * a32 + (q >> 1), b16 = 2 * b32 + (q & 1);
            if (idx == NM16 - BAR16) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!(X3_ABLATE & 8)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4a c = {acc[a32][b32][4 * q], acc[a32][b32][4 * q + 1], acc[a32][b32][4 * q + 2], acc[a32][b32][4 * q + 3]};
            if (FIRST && t == 0) {                                 // the pending stores of the 32 x 32 block, then its restart from zero
                if (q == 0) {
                    if (STG) {
                        if (sb > 0) flush_block((sb - 1) / TN, (sb - 1) % TN);
                        stage_block(a32, b32);
                    } else if (!ATOMIC) {
                        store_block(a32, b32);
                    }
                }
                c = (f32x4a){0.f, 0.f, 0.f, 0.f};
            }
            if (FIRST && STG && idx == PM) flush_block(TM - 1, TN - 1);
            const X3Parts &A = fa[0][a16], &W = fw[0][b16];
            const u32x4 ap = t <= 2 ? A.h : (t == 4 ? A.l : A.m);
            const u32x4 wp = (t == 0 || t == 3 || t == 4) ? W.h : (t == 2 ? W.l : W.m);
            c = ATOMIC ? x3_mfma16(ap, wp, c) : x3_mfma16(wp, ap, c);
            acc[a32][b32][4 * q] = c[0]; acc[a32][b32][4 * q + 1] = c[1]; acc[a32][b32][4 * q + 2] = c[2]; acc[a32][b32][4 * q + 3] = c[3];
            if (idx % X3_FENCE16 == 0) __builtin_amdgcn_sched_barrier(0);
            if (!(X3_ABLATE & 16)) {
                if (idx >= 1 && idx < 1 + SP * L1 && (idx - 1) % SP == 0) {
                    const int j = (idx - 1) / SP;
                    if (j < FW) read_part(st, false, 1, j);                               // Wm
                    else if (j < 2 * FW) read_part(st, false, 2, j - FW);                 // Wl
                    else if (j < 2 * FW + FA) read_part(st, true, 1, j - 2 * FW);         // Am
                    else read_part(st, true, 2, j - 2 * FW - FA);                         // Al
                }
                const int ja = idx - (NM16 - BAR16 + 1), jw = idx - 5 * PM;
                if (ja >= 0 && ja < 2 * FA && ja % 2 == 0) read_part(st ^ 1, true, 0, ja / 2);      // Ah of the next chunk
                if (jw >= 0 && jw < 2 * FW && jw % 2 == 0) read_part(st ^ 1, false, 0, jw / 2);     // Wh of the next chunk
            }
            if (idx >= C_LO && idx < C_HI16) {
                const int i0 = idx - C_LO, span = C_HI16 - C_LO;
                const int k0 = (i0 * NCT) / span, k1 = ((i0 + 1) * NCT) / span;
#pragma unroll
                for (int k = k0; k < k1; ++k) conv_task(st ^ 1, k);
            }
        }
    };

    // ---- prologue: chunk 0 converted into stage 0, chunk 1 in the raw registers, the first fragments read
    // NP = 2: the un-scales of the compute cursor's item -- its rows' and columns' maxima are fetched at the top of the item's first
    // chunk, one entry (or two) per thread, and written to the table in LDS behind that chunk (behind its barrier: every wave has
    // left the previous item's finish_item by then)
    constexpr int NUN = NP == 2 ? (BM + BN + NTH - 1) / NTH : 1;
    unsigned unb[NUN];
    auto fetch_unscales = [&]() {
        if (NP != 2) return;
        int tm, tn;
        decode(cp.tile, tm, tn);
        const long long m0 = (long long)tm * BM;
        const int n0 = tn * BN;
        const int mrows = (int)min((long long)BM, p.M - m0), ncols = min(BN, p.N - n0);
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void *)(p.max_a + m0), 0, mrows * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void *)(p.max_w + n0), 0, ncols * 4, 0x00020000);
        int lane_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_));
#pragma unroll
        for (int i = 0; i < NUN; ++i) {
            const int e = wave * 64 + lane_ + i * NTH;
            unb[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rA, e < BM ? (unsigned)e * 4u : NT_OOB, 0, 0) |
                     (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rW, e >= BM ? (unsigned)(e - BM) * 4u : NT_OOB, 0, 0);
        }
    };
    auto store_unscales = [&]() {
        if (NP != 2) return;
        int lane_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_));
#pragma unroll
        for (int i = 0; i < NUN; ++i) {
            const int e = wave * 64 + lane_ + i * NTH;
            if (e < BM + BN) untab[e] = x2_unscale(unb[i]);
        }
    };

    issue_begin();
    fetch_maxima();                                                // (the first item's scales)
    adopt_scales();
#pragma unroll
    for (int q = 0; q < NQ; ++q) load_quad(q);
    advance_load();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        conv_pair(q, 0);
        conv_pair(q, 1);
        conv_write(0, q);
    }
    if (NP == 2 && ld.valid && ld.kc == ld.kb) {                   // (a one-chunk item: the second chunk of the sequence opens the next one)
        fetch_maxima();
        adopt_scales();
    }
    issue_begin();
#pragma unroll
    for (int q = 0; q < NQ; ++q) load_quad(q);
    advance_load();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (MS == 32) {
#pragma unroll
        for (int r = 0; r < NFR; ++r) read_frag(0, 0, 0, r);
    } else {
#pragma unroll
        for (int f = 0; f < FA; ++f) read_part(0, true, 0, f);
#pragma unroll
        for (int f = 0; f < FW; ++f) read_part(0, false, 0, f);
    }
    int stage = 0;
    while (cp.valid) {
        // the item's first chunk is peeled off the loop: one code path per loop body, so the loop-carried registers (raw
        // values in flight, accumulators) need no copies at a merge
        // NP = 2: when the chunk about to be loaded opens an item, that item's row maxima are fetched now and become the conversion's
        // scales behind this chunk (whose conversion tasks still split the item before)
        const bool open0 = NP == 2 && ld.valid && ld.kc == ld.kb;
        if (open0) fetch_maxima();
        fetch_unscales();
        issue_begin();                                             // the chunk the conversion tasks reload the registers with
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MS == 32) chunk(std::true_type(), stage);
        else chunk16(std::true_type(), stage);
        __builtin_amdgcn_sched_barrier(0);
        if (open0) adopt_scales();
        store_unscales();
        advance_load();
        stage ^= 1;
        cp.kc++;
        while (cp.kc != cp.ke) {
            const bool open1 = NP == 2 && ld.valid && ld.kc == ld.kb;
            if (open1) fetch_maxima();
            issue_begin();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MS == 32) chunk(std::false_type(), stage);
            else chunk16(std::false_type(), stage);
            __builtin_amdgcn_sched_barrier(0);
            if (open1) adopt_scales();
            advance_load();
            stage ^= 1;
            cp.kc++;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NP == 2 && cp.ke - cp.kb == 1) {                       // (a one-chunk item: its table was written behind its only barrier)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
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

#ifndef X3_KERNEL_ONLY                // (tools/x3_inst.hip compiles single instances of the kernel for ISA inspection)
// ------------------------------------------------------------------ host side
static bool x3_shape16(int cfg_index, bool atomic, bool epi, bool pw);

// ---- stream-K tails without atomics (round 5).  A launch whose tiles do not fill the last round of workgroups used to finish with
// a second launch that split the leftover tiles' k range over all CUs and ADDED its partial tiles into C with fp32 atomics (after
// a zero-fill of those rows).  Inside the iteration those tails were the slowest launches per flop of the whole step -- conv2's
// forward: 371 us for 0.19 rounds of work that take 77 us as whole tiles (tools/sk_tails.py: 1.6 ms of tails per iteration on the
// issuing stream).  Now the tail's workgroups store their partial tiles -- whole BM x BN tiles, the ordinary store path -- into a
// workspace and a small kernel adds a tile's slices (in order: deterministic) plus bias / addend into C: no atomics, no zero-fill.
// The workspace is the CALLER's (kernels never allocate, and an allocation under a stream capture would end the capture):
// pdgn_gemm_tail_workspace_floats says how much a problem's tail wants, pdgn_gemm_set_tail_workspace hands a buffer to the NEXT
// contraction call of the calling thread (thread-local, consumed by that call); without one the atomic form runs.
static thread_local float *x3_tail_ws = nullptr;            // handed in for the NEXT contraction call (pdgn_gemm_set_tail_workspace)
static thread_local long long x3_tail_ws_floats = 0;
static thread_local float *x3_cur_ws;                       // ... taken into the call in progress (X3Handover, below)
static thread_local long long x3_cur_ws_floats;
static float *x3_take_workspace(size_t floats) {
    static const bool off = [] { const char *e = getenv("PDGN_X3_SK_WS"); return e && e[0] == '0'; }();   // 0: the atomic tails (A/B)
    float *p = (!off && x3_cur_ws && (size_t)x3_cur_ws_floats >= floats) ? x3_cur_ws : nullptr;
    x3_cur_ws = nullptr;
    x3_cur_ws_floats = 0;
    return p;
}

// C tile t of the tail = sum over its parts in the workspace (+ bias + addend); gridDim.y workgroups per tile (BM % (8 * 1) == 0).
// Aligned form (seg == 0): the S slices ws[s T + t].  Flattened form (seg > 0; per = iterations per workgroup, KC = chunks per
// tile): the workgroups v0 .. v1 whose iteration ranges meet the tile's, each one's part at slot v seg + (t - its first tile) --
// in workgroup order: a fixed order of summation either way.
__global__ __launch_bounds__(256) void x3_sk_reduce_kernel(const float *__restrict__ ws, int S, int T, int BM, int BN, int tile_begin,
                                                           int tiles_m, int tiles_n, long long M, int N, float *__restrict__ C, int ldc,
                                                           const float *__restrict__ bias, const float *__restrict__ addend, int ldadd,
                                                           int seg, long long per, int KC) {
    const int t = blockIdx.x, tile = tile_begin + t;
    const int per_group = NT_GROUP_M * tiles_n;
    const int grp = tile / per_group, r = tile - grp * per_group, first = grp * NT_GROUP_M;
    const int gsz = min(NT_GROUP_M, tiles_m - first);
    const int tn = r / gsz, tm = first + (r - tn * gsz);
    const long long m0 = (long long)tm * BM;
    const int n0 = tn * BN;
    const int mrows = (int)min((long long)BM, M - m0), ncols = min(BN, N - n0);
    const int q4 = BN / 4;
    const int rows_per = BM / (int)gridDim.y;                     // gridDim.y workgroups share a tile by rows (48 tiles alone fill a fifth of the chip)
    const int ne = rows_per * q4;                                  // float4 elements of this workgroup
    // few elements per workgroup (few tiles, many slices: a 256 x 128 weight gradient summed over 128 slices): G threads share an
    // element, thread g takes the slices g, g + G, ... in order and the G partial sums are added in g order -- a fixed order still
    const int G = (!seg && ne < 256) ? 256 / ne : 1;
    __shared__ float4 red[256];
    for (int e0 = 0; e0 < ne; e0 += 256 / G) {
        const int e = e0 + (int)threadIdx.x % (256 / G), g = (int)threadIdx.x / (256 / G);
        const int row = blockIdx.y * rows_per + e / q4, c4 = (e % q4) * 4;
        const bool in = e < ne && row < mrows && c4 < ncols;       // (N % 4 == 0: a float4 is all in or all out)
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in && !seg) {
            bool have = false;
#pragma unroll 4                                                   // (several slices' loads in flight; the additions keep their order)
            for (int sl = g; sl < S; sl += G) {
                const float4 b = *reinterpret_cast<const float4 *>(ws + (((size_t)sl * T + t) * BM + row) * BN + c4);
                if (!have) { a = b; have = true; }
                else { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            }
        } else if (in) {
            const long long f0 = (long long)t * KC, f1 = f0 + KC - 1;
            const long long v0 = f0 / per, v1 = f1 / per;
            for (long long vv = v0; vv <= v1; ++vv) {
                const size_t slot = (size_t)(vv * seg + (t - (vv * per) / KC));
                const float4 b = *reinterpret_cast<const float4 *>(ws + (slot * BM + row) * BN + c4);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
        }
        if (G > 1) {                                               // (uniform over the workgroup)
            red[threadIdx.x] = a;
            __syncthreads();
            if (g == 0)
                for (int gg = 1; gg < G; ++gg) {
                    const float4 b = red[gg * (256 / G) + (int)threadIdx.x];
                    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                }
            __syncthreads();
        }
        if (!in || g != 0) continue;
        if (bias) { const float4 b = *reinterpret_cast<const float4 *>(bias + n0 + c4); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        if (addend) {
            const float4 b = *reinterpret_cast<const float4 *>(addend + (m0 + row) * ldadd + n0 + c4);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *reinterpret_cast<float4 *>(C + (m0 + row) * ldc + n0 + c4) = a;
    }
}
// ---- two-part mode (NP = 2): the operands' power-of-two scales, ONE PER ROW of each operand as the kernel sees it (round 6: per
// operand until then).  fp16 holds 2^-14 .. 65504 (2^-24 with subnormals): row r is multiplied by 2^e_r, e_r = 14 - floor(log2(max
// |x| over the row)), before its split -- max 2^e in [2^14, 2^15): nothing overflows, and a value keeps its full 22 bits while it
// is within 2^-16 of ITS ROW's largest (smaller ones lose bits to fp16's subnormal spacing: an absolute error of at most 2^-39 of
// the row's maximum).  C[m, n] leaves the scaled domain by 2^-e_A[m] 2^-e_W[n] in the epilogue (exact).  For an operand given
// transposed (the input gradient's weight, both operands of a weight gradient) the kernel's rows are the COLUMNS of the matrix in
// memory: a weight gradient dW[n, k] = sum_r dY[r, n] X[r, k] scales dY per column n and X per column k.
// The maxima (bit patterns of |x|: unsigned order = magnitude order, so they combine by integer max in any order) come from
// x2_maxima_kernel (one pass at memory speed: row maxima and / or column maxima), from the caller (pdgn_gemm_set_operand_scales),
// or from the kernel that wrote the operand; a pre-split operand carries its rows' maxima behind its planes.  The library's own
// scans take their arrays from an arena the CALLER provides once (pdgn_gemm_set_scale_slots: the library never allocates; it is
// used round-robin, so it must hold the arrays of all launches that can be in flight).
template <int NU>
__global__ __launch_bounds__(1024) void x2_maxima_kernel(const float *__restrict__ X, long long rows, int cols, int ld, int slab_units,
                                                         long long rows_per_wg, unsigned *__restrict__ rowmax,
                                                         unsigned *__restrict__ colmax, int row_atomic, int upr) {
    // a workgroup takes rows_per_wg consecutive rows of one slab of 64 NU 16-B column units; a WAVE takes every 16th of them, lane l
    // the units l, l + 64, ... of the slab: its columns are the same for every row (running column maxima in registers).  Narrow
    // matrices (NU = 1, upr = 2 .. 32 units per row, a power of two): 64 / upr rows per wave and pass, lane l = (row l / upr, unit
    // l % upr) -- a 64-column operand read one row per wave and pass ran at 1.8 TB/s
    extern __shared__ unsigned x2_cmax[];                          // [4 slab_units] when colmax
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u0 = blockIdx.y * slab_units, nu = min(slab_units, cols / 4 - u0);
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    if (colmax) {
        for (int i = threadIdx.x; i < 4 * nu; i += 1024) x2_cmax[i] = 0u;
        __syncthreads();
    }
    const int G = 64 / upr, ul = lane & (upr - 1), gl = lane / upr;     // (upr = 64: G = 1, ul = lane, gl = 0)
    uint4 cm[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) cm[i] = make_uint4(0u, 0u, 0u, 0u);
    for (long long r = r0 + (long long)wave * G + gl; r < r1; r += 16 * G) {
        const uint4 *__restrict__ P = reinterpret_cast<const uint4 *>(X + r * ld) + u0;
        uint4 v[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) v[i] = ul + 64 * i < nu ? P[ul + 64 * i] : make_uint4(0u, 0u, 0u, 0u);
        unsigned rm = 0u;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            v[i].x &= 0x7fffffffu; v[i].y &= 0x7fffffffu; v[i].z &= 0x7fffffffu; v[i].w &= 0x7fffffffu;
            rm = max(rm, max(max(v[i].x, v[i].y), max(v[i].z, v[i].w)));
            cm[i].x = max(cm[i].x, v[i].x); cm[i].y = max(cm[i].y, v[i].y); cm[i].z = max(cm[i].z, v[i].z); cm[i].w = max(cm[i].w, v[i].w);
        }
        if (rowmax) {
            for (int o = upr >> 1; o >= 1; o >>= 1) rm = max(rm, (unsigned)__shfl_xor((int)rm, o));
            if (ul == 0) {
                if (row_atomic) atomicMax(rowmax + r, rm);         // (several slabs: the launcher zero-filled the array)
                else rowmax[r] = rm;
            }
        }
    }
    if (!colmax) return;
#pragma unroll
    for (int i = 0; i < NU; ++i)
        if (ul + 64 * i < nu) {
            unsigned *c = x2_cmax + 4 * (ul + 64 * i);
            atomicMax(c, cm[i].x); atomicMax(c + 1, cm[i].y); atomicMax(c + 2, cm[i].z); atomicMax(c + 3, cm[i].w);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * nu; i += 1024) atomicMax(colmax + 4 * u0 + i, x2_cmax[i]);
}
// rowmax[rows] and / or colmax[cols] of |X| (rows x cols, pitch ld; cols, ld multiples of 4, X 16-byte aligned) on stream s
static int x2_maxima_launch(const float *X, long long rows, int cols, int ld, unsigned *rowmax, unsigned *colmax, hipStream_t s) {
    if (rows < 1 || cols < 4 || cols % 4 || ld % 4 || ld < cols || ((uintptr_t)X & 15) || (!rowmax && !colmax)) return PDGN_ERR_INVALID;
    const int units = cols / 4;
    // units per lane: 1, 2, 4 or 8 (2048 columns per slab: 64 registers of values and running column maxima); wider matrices in several slabs
    const int per = (units + 63) / 64;
    const int NU = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : 8;
    const int slab = 64 * NU, slabs = (units + slab - 1) / slab;
    const int cus = nt_cus();
    int upr = 64;                                                  // NU = 1: units per row rounded up to a power of two (narrow matrices: several rows per wave)
    if (NU == 1) { upr = 1; while (upr < units) upr *= 2; }
    long long wgs = (rows + 16 * (64 / upr) - 1) / (16 * (64 / upr));      // at least one pass per wave
    const long long cap = colmax ? (long long)cus / slabs + 1 : 8LL * cus;    // column maxima: every workgroup ends with 4 slab atomics per column unit
    wgs = wgs < 1 ? 1 : (wgs > cap ? cap : wgs);
    const int rq = 16 * (64 / upr);                                // rows a workgroup takes per pass
    const long long rpw = ((rows + wgs - 1) / wgs + rq - 1) / rq * rq;
    wgs = (rows + rpw - 1) / rpw;
    if (wgs > 0x7fffffffLL || slabs > 65535) return PDGN_ERR_INVALID;
    const int row_atomic = rowmax && slabs > 1;
    if (row_atomic && hipMemsetAsync(rowmax, 0, (size_t)rows * 4, s) != hipSuccess) return pdgn_launch_status();
    if (colmax && hipMemsetAsync(colmax, 0, (size_t)cols * 4, s) != hipSuccess) return pdgn_launch_status();
    const size_t lds = colmax ? (size_t)slab * 16 : 0;
#define X2_MAX_CALL(N_)                                                                                                         \
    hipLaunchKernelGGL((x2_maxima_kernel<N_>), dim3((unsigned)wgs, (unsigned)slabs), dim3(1024), lds, s, X, rows, cols, ld, slab, rpw, \
                       rowmax, colmax, row_atomic, upr)
    switch (NU) {
        case 1: X2_MAX_CALL(1); break;
        case 2: X2_MAX_CALL(2); break;
        case 4: X2_MAX_CALL(4); break;
        default: X2_MAX_CALL(8); break;
    }
#undef X2_MAX_CALL
    return pdgn_launch_status();
}
int x2_maxima_launch_ext(const float *X, long long rows, int cols, int ld, unsigned *rowmax, unsigned *colmax, hipStream_t s) {      // (split.hip)
    return x2_maxima_launch(X, rows, cols, ld, rowmax, colmax, s);
}
// the caller's arena for the library's own scans: bytes handed out round-robin in 1-KB units
static std::atomic<unsigned *> x2_ring{nullptr};
static std::atomic<unsigned> x2_ring_slots{0};
static unsigned *x2_arena_take(long long entries) {
    static std::atomic<unsigned long long> next{0};
    unsigned *ring = x2_ring.load();
    const unsigned slots = x2_ring_slots.load();
    const unsigned long long need = (unsigned long long)(entries + 255) / 256;
    if (!ring || !slots || need > slots) return nullptr;
    for (;;) {
        const unsigned long long at = next.fetch_add(need);
        const unsigned long long pos = at % slots;
        if (pos + need <= slots) return ring + (size_t)pos * 256;  // (a range that would wrap is skipped: the next take starts over)
    }
}
// the maxima of the kernel-side rows of an operand (rows x cols in memory, pitch ld) on stream s: of its rows, or -- bycol: the
// operand is given transposed -- of its columns.  A device pointer valid for the launches that follow on s (NULL: no arena)
const unsigned *x2_scan(const float *X, long long rows, int cols, int ld, bool bycol, hipStream_t s) {
    unsigned *out = x2_arena_take(bycol ? cols : rows);
    if (!out) return nullptr;
    if (x2_maxima_launch(X, rows, cols, ld, bycol ? nullptr : out, bycol ? out : nullptr, s) != 0) return nullptr;
    return out;
}
static thread_local const unsigned *x2_next_max_a = nullptr, *x2_next_max_w = nullptr;     // handed in for the NEXT contraction call's operands

// What the caller handed over for "the next contraction call" (a tail workspace, operand maxima) belongs to THAT call, whether it
// launches anything or returns an error first: every public entry point opens with an X3Handover, which takes the pending
// hand-overs into the call (x3_cur_*: what launch() consumes) and leaves nothing behind for a later call to pick up by mistake.
static thread_local const unsigned *x2_cur_max_a = nullptr, *x2_cur_max_w = nullptr;
struct X3Handover {
    X3Handover() {
        x3_cur_ws = x3_tail_ws; x3_cur_ws_floats = x3_tail_ws_floats;
        x2_cur_max_a = x2_next_max_a; x2_cur_max_w = x2_next_max_w;
        x3_tail_ws = nullptr; x3_tail_ws_floats = 0;
        x2_next_max_a = x2_next_max_w = nullptr;
    }
    ~X3Handover() {
        x3_cur_ws = nullptr; x3_cur_ws_floats = 0;
        x2_cur_max_a = x2_cur_max_w = nullptr;
    }
};

// workgroups per tile of x3_sk_reduce_kernel: 8, more when there are few tiles (a weight gradient of 2 tiles in 128 slices is 128
// dependent loads per thread: 64 us with 16 workgroups)
static int x3_reduce_split(int tiles) {
    int y = 8;
    while (tiles * y < 512 && y < 64) y *= 2;
    return y;
}

// gemm_rp.hip: the row-panel kernel of the short-reduction, wide-result products on two-part planes
bool rp_takes(long long m, int n, int k);
int rp_launch(long long m, int n, int k, const float *A, int lda, const unsigned short *Wp, int ldw, long long wplane, float *C, int ldc,
              float *stat_part, hipStream_t s);
// gemm_x3_16.hip: instance (tile cfg, flags = ATOMIC | WT << 1 | AT << 2 | EPI << 3 | PW << 4) of gemm_x3_kernel<..., 16>
void x3_launch16(int cfg, int flags, int grid, hipStream_t s, const NtArgs &a);
// gemm_x3_h2.hip: the same of gemm_x3_kernel<..., 32, 2> (two fp16 parts), and an instance's host symbol
void x3_launch_h2(int flags, int grid, hipStream_t s, const NtArgs &a, bool eight_waves);
const void *x3_symbol_h2(int flags, bool eight_waves);
// Where the two-part form pays (measured, tools/x2_shapes.py: every contraction of an iteration alone in both forms): it halves
// the matrix-core work of a launch and needs the operands' maxima first -- a pass over every operand that does not bring them
// along.  On the 256 x 128 tile, with a reduction of at least 128 and >= 20 GFLOP, the products are the larger part of the
// launch (conv2's dense half 927 -> 782 us with both scans, its input gradient 952 -> 693, the per-point product 761 -> 610);
// the smaller tiles' launches are bound by their operand path and only pay the scans (+5 .. 100 us each).  A launch that
// would have to read more than 4.5 bytes per kflop for the maxima keeps three parts (the per-point product's input gradient:
// 1.8 GB of dY for 118 GFLOP).
static bool x2_pays(int cfg, long long m, int n, int k, long long scan_bytes) {
    if (nt_switches().mode != 2 || cfg != 0 || k < 128 || m >= (1LL << 28)) return false;
    const double flops = 2.0 * (double)m * n * k;
    static const double min_flops = [] { const char *e = getenv("PDGN_X2_MIN_GFLOP"); return (e && *e) ? atof(e) * 1e9 : 2e10; }();   // (measurement)
    return flops >= min_flops && (double)scan_bytes <= 4.5e-3 * flops;
}

template <int TM, int TN, int WM, int WN, int OCC, int RATE, int CFG>      // RATE: fp32-equivalent kflop / us a CU sustains on this tile's loop
struct X3Cfg {
    static constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    static constexpr int LDS = 2 * 3 * (BM + BN) * 64;              // two stages of three bf16 parts of [rows][32 k]
    static constexpr int WG_PER_CU = OCC;

    struct Plan {
        int tiles_m, tiles_n, kchunks, grid_dp, dp_tiles, grid_sk;
        long long sk_per_wg;
        double cost;                                               // launch model, microseconds
    };
    // Launch model as in gemm_nt.hip, with this loop's measured rate (tools/x3_check.py on MI355X).
    static int dp_grid(long long tiles, int slots, int kc) {
        int tpw = (64 + kc - 1) / kc;
        tpw = tpw < 1 ? 1 : (tpw > 8 ? 8 : tpw);
        const long long g = (tiles + tpw - 1) / tpw;
        const long long lo = tiles < slots ? tiles : slots;
        return (int)(g < lo ? lo : g);
    }
    static double chunk_us(int w) { return 2.0 * BM * BN * NT_BK * w / (RATE * 1e3); }
    static Plan plan(long long m, int n, int k, bool allow_sk) {
        Plan pl;
        pl.tiles_m = cdiv(m, BM);
        pl.tiles_n = cdiv(n, BN);
        pl.kchunks = cdiv(k, NT_BK);
        const long long T = (long long)pl.tiles_m * pl.tiles_n;
        const int cus = nt_cus(), slots = cus * WG_PER_CU, KC = pl.kchunks;
        const long long rounds = T / slots, tail = T - rounds * slots;
        const double ov = 2.0;
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

    template <bool ATOMIC, bool WT, bool AT, bool EPI, bool PW = false>
    static void go(int grid, hipStream_t s, const NtArgs &a, bool two_part = false) {
        // the matrix instruction is chosen per instance class (tile, atomic / extended epilogue / pre-split operand): nt_switches().shape16
        // (the 16x16x32 instances live in their own translation unit, gemm_x3_16.hip: the two compile side by side)
        const int flags = (ATOMIC ? 1 : 0) | (WT ? 2 : 0) | (AT ? 4 : 0) | (EPI ? 8 : 0) | (PW ? 16 : 0);
        if (two_part) x3_launch_h2(flags, grid, s, a, a.stat_part == nullptr);      // (256 x 128 tiles only: x2_pays; eight waves unless BatchNorm partials are emitted)
        else if (x3_shape16(CFG, ATOMIC, EPI, PW)) x3_launch16(CFG, flags, grid, s, a);
        else hipLaunchKernelGGL((gemm_x3_kernel<TM, TN, WM, WN, OCC, ATOMIC, WT, AT, EPI, PW, 32>), dim3(grid), dim3(64 * WM * WN), 0, s, a);
    }

    // How a stream-K tail (the leftover tiles' k ranges over all CUs) is laid out when its partial tiles go to a workspace.
    // Aligned: every tile's chunks in s_tail equal slices, workgroup v = (slice, tile): the workgroups that run together read the
    // same chunks of different tiles.  When the tiles do not divide the CUs well (140 leftover tiles on 256 CUs: one slice each
    // = 55 % of the chip, the per-point product's input gradient) the FLATTENED order instead: iteration ranges of equal length,
    // a workgroup's range may span seg tiles.
    struct Tail {
        int t_tail, s_tail, per_tail, grid, seg;
        long long per_flat, floats;
    };
    static Tail tail_layout(const Plan &pl) {
        Tail t = {0, 0, 0, 0, 0, 0, 0};
        if (!pl.grid_sk) return t;
        const long long T = (long long)pl.tiles_m * pl.tiles_n;
        const int slots_ = nt_cus() * WG_PER_CU;
        t.t_tail = (int)(T - pl.dp_tiles);
        t.s_tail = slots_ / t.t_tail < 1 ? 1 : slots_ / t.t_tail;
        t.s_tail = t.s_tail > pl.kchunks ? pl.kchunks : t.s_tail;
        t.per_tail = (pl.kchunks + t.s_tail - 1) / t.s_tail;
        t.s_tail = (pl.kchunks + t.per_tail - 1) / t.per_tail;
        t.grid = t.s_tail * t.t_tail;
        const long long iters = (long long)t.t_tail * pl.kchunks;
        if (t.grid * 5 < slots_ * 4 && iters >= (long long)slots_ * 8) {     // under 80 % of the chip, and ranges of >= 8 chunks to hand out
            t.per_flat = (iters + slots_ - 1) / slots_;
            t.grid = (int)((iters + t.per_flat - 1) / t.per_flat);
            t.seg = (int)((t.per_flat + pl.kchunks - 2) / pl.kchunks) + 1;
            t.floats = (long long)t.grid * t.seg * BM * BN;
        } else {
            t.floats = (long long)t.grid * BM * BN;
        }
        return t;
    }
    // floats of workspace the stream-K tail of (m, n, k) wants (0: no tail)
    static long long tail_floats(long long m, int n, int k, bool allow_sk) { return tail_layout(plan(m, n, k, allow_sk)).floats; }
    // weight gradients (split-K over few output tiles): the number of NON-EMPTY k slices per tile, and the workspace their partial
    // tiles want (0: the launch is not a split-K one)
    static int at_slices(const Plan &pl) {
        const long long T = (long long)pl.tiles_m * pl.tiles_n;
        const int slots = nt_cus() * WG_PER_CU;
        const int S = (int)(slots / T) < pl.kchunks ? (int)(slots / T) : pl.kchunks;
        const int per = (pl.kchunks + S - 1) / S;
        return (pl.kchunks + per - 1) / per;
    }
    static long long at_floats(long long m, int n, int k) {
        const Plan pl = plan(m, n, k, true);
        const long long T = (long long)pl.tiles_m * pl.tiles_n;
        if (T > nt_cus() * WG_PER_CU || !nt_switches().splitk) return 0;
        return (long long)at_slices(pl) * T * BM * BN;
    }

    template <bool WT, bool AT = false>
    static int launch(long long m, int n, int k, const float *A, int lda, const float *W, int ldw, const float *bias,
                      const float *addend, int ldadd, float *C, int ldc, float *stat_part, hipStream_t s,
                      const NtEpi &epi = NtEpi(), bool prezeroed = false, const unsigned short *Wp = nullptr, long long wplane = 0,
                      int parts = 3) {
        const bool allow_sk = stat_part == nullptr && ldc == n && !epi.any();
        const Plan pl = plan(m, n, k, allow_sk);
        NtArgs a;
        a.M = m; a.N = n; a.K = k; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldadd = ldadd;
        a.A = A; a.W = W; a.bias = bias; a.addend = addend; a.C = C; a.stat_part = stat_part;
        a.row_bias = epi.row_bias; a.ld_rb = epi.ld_rb; a.rows_per_group = epi.rows_per_group > 0 ? epi.rows_per_group : 1;
        a.rpg_magic = a.rows_per_group > 1 ? (unsigned)(0x100000000ULL / (unsigned)a.rows_per_group) + 1u : 0u;
        a.rb_bytes = epi.row_bias ? (int)((((m + a.rows_per_group - 1) / a.rows_per_group - 1) * (long long)epi.ld_rb + n) * 4) : 0;
        a.act = epi.act; a.gate = epi.gate; a.ldgate = epi.ldgate; a.sk_split = 0;
        a.Wp = Wp; a.wplane = wplane;
        a.sk_ws = nullptr;
        a.sk_seg = 0;
        a.max_a = a.max_w = nullptr;
        // two parts or three: pre-split planes say it themselves; otherwise where it pays (x2_pays: mode 2, this tile, enough
        // matrix-core work per byte that still has to be scanned for its maximum)
        const unsigned *hand_a = x2_cur_max_a, *hand_w = x2_cur_max_w;
        x2_cur_max_a = x2_cur_max_w = nullptr;
        const bool two = Wp ? parts == 2
                            : x2_pays(CFG, m, n, k, (hand_a ? 0 : (long long)m * k * 4) + (hand_w ? 0 : (long long)n * k * 4));
        if (two && CFG != 0) return PDGN_ERR_INVALID;              // (two-part planes for a problem the other tiles take: x2_pays said no)
        if (two) {
            // per-row maxima of each operand as the kernel sees it (a transposed operand: the columns of the matrix in memory)
            a.max_a = hand_a ? hand_a : x2_scan(A, AT ? (long long)k : m, AT ? (int)m : k, lda, AT, s);
            if (Wp) a.max_w = reinterpret_cast<const unsigned *>(Wp + 2 * wplane);      // (behind the two planes: pdgn_split_f16x2)
            else a.max_w = hand_w ? hand_w : x2_scan(W, WT ? k : n, WT ? n : k, ldw, WT, s);
            if (!a.max_a || !a.max_w) return PDGN_ERR_INVALID;     // (no arena: pdgn_gemm_set_scale_slots)
        }
        a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n; a.kchunks = pl.kchunks;
        a.dbg = 0;
#ifdef PDGN_NT_DEBUG
        { const char *e = getenv("PDGN_NT_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
        const long long T = (long long)pl.tiles_m * pl.tiles_n;
        if (AT && T <= nt_cus() * WG_PER_CU && nt_switches().splitk) {
            // weight gradients: few output tiles, a reduction of 10^4 .. 10^5.5 rows.  Split-K: the workgroups that run at the same
            // time work on the SAME row range of different tiles (tile = v % T), so the XCD's L2 serves the operand panels they
            // share; the flattened stream-K order gives neighbouring workgroups neighbouring row ranges of one tile and every
            // panel is fetched from HBM once per tile (conv2's dense half: 4.4 GB for 0.8 GB of operands).
            const int slots = nt_cus() * WG_PER_CU;
            const int S = (int)(slots / T) < pl.kchunks ? (int)(slots / T) : pl.kchunks;
            a.tile_begin = 0; a.tile_end = (int)T; a.sk_split = (int)T; a.sk_per_wg = (pl.kchunks + S - 1) / S;
            // with a workspace (pdgn_gemm_tn_big_workspace_floats): the slices' partial tiles go there and x3_sk_reduce_kernel sums
            // them in slice order -- no atomics, no zero-fill, the same bits every run; without one: fp32 atomics into a zeroed dW
            const int s_ws = at_slices(pl);
            float *ws = x3_take_workspace((size_t)s_ws * T * BM * BN);
            if (ws) {
                a.sk_ws = ws;
                go<false, WT, AT, false>((int)T * s_ws, s, a, two);
                hipLaunchKernelGGL(x3_sk_reduce_kernel, dim3((int)T, x3_reduce_split((int)T)), dim3(256), 0, s, ws, s_ws, (int)T, BM, BN, 0, pl.tiles_m, pl.tiles_n, m, n,
                                   C, ldc, nullptr, nullptr, 0, 0, 0LL, pl.kchunks);
                return pdgn_launch_status();
            }
            if (!prezeroed && hipMemsetAsync(C, 0, (size_t)m * ldc * sizeof(float), s) != hipSuccess) return pdgn_launch_status();
            go<true, WT, AT, false>((int)T * S, s, a, two);
            return pdgn_launch_status();
        }
        // stream-K tail: the leftover tiles' k range in S slices, partial tiles to the workspace + a reduce (no atomics); the atomic
        // form (zero-fill + fp32 atomics into C) when there is no workspace
        const Tail tl = tail_layout(pl);
        float *ws = (tl.t_tail > 0 && !AT) ? x3_take_workspace((size_t)tl.floats) : nullptr;
        if (pl.grid_sk && !ws) {
            const long long r0 = (long long)(pl.dp_tiles / (NT_GROUP_M * pl.tiles_n)) * NT_GROUP_M * BM;
            if (!prezeroed && hipMemsetAsync(C + r0 * ldc, 0, (size_t)(m - r0) * ldc * sizeof(float), s) != hipSuccess) return pdgn_launch_status();
        }
        constexpr bool CAN_PW = !WT && !AT;
        if (pl.grid_dp) {
            a.tile_begin = 0; a.tile_end = pl.dp_tiles; a.sk_per_wg = 0;
            if (CAN_PW && Wp) {
                if (epi.any()) go<false, false, false, true, CAN_PW>(pl.grid_dp, s, a, two);
                else go<false, false, false, false, CAN_PW>(pl.grid_dp, s, a, two);
            } else if (!AT && epi.any()) go<false, WT, false, true>(pl.grid_dp, s, a, two);
            else go<false, WT, AT, false>(pl.grid_dp, s, a, two);
        }
        if (pl.grid_sk && ws) {
            a.tile_begin = pl.dp_tiles; a.tile_end = (int)T; a.sk_ws = ws;
            a.sk_split = tl.seg ? 0 : tl.t_tail; a.sk_seg = tl.seg; a.sk_per_wg = tl.seg ? tl.per_flat : tl.per_tail;
            a.bias = nullptr; a.addend = nullptr;                  // (the reduce adds them)
            if (CAN_PW && Wp) go<false, false, false, false, CAN_PW>(tl.grid, s, a, two);
            else go<false, WT, AT, false>(tl.grid, s, a, two);
            hipLaunchKernelGGL(x3_sk_reduce_kernel, dim3(tl.t_tail, x3_reduce_split(tl.t_tail)), dim3(256), 0, s, ws, tl.s_tail, tl.t_tail, BM, BN, pl.dp_tiles, pl.tiles_m,
                               pl.tiles_n, m, n, C, ldc, bias, addend, ldadd, tl.seg, tl.per_flat, pl.kchunks);
        } else if (pl.grid_sk) {
            a.tile_begin = pl.dp_tiles; a.tile_end = (int)T; a.sk_per_wg = pl.sk_per_wg;
            if (CAN_PW && Wp) go<true, false, false, false, CAN_PW>(pl.grid_sk, s, a, two);
            else go<true, WT, AT, false>(pl.grid_sk, s, a, two);
        }
        return pdgn_launch_status();
    }
};

typedef X3Cfg<2, 2, 2, 2, 1, 660, 1> X3Square;   // 128 x 128, 4 waves of 64 x 64 (96 KB of LDS): one per CU
typedef X3Cfg<4, 2, 2, 2, 1, 760, 0> X3Big;      // 256 x 128, 4 waves of 128 x 64 (144 KB of LDS): one per CU, one wave per SIMD
// (the bf16 form as eight waves of 64 x 64 -- what the two-part form runs on, gemm_x3_h2.hip -- has no room for the result staging
// (147 + 32 KB of LDS) and measured 1055 -> 1027 us on conv2's dense half, 860 -> 905 on the per-point product: not built)
typedef X3Cfg<2, 1, 2, 2, 2, 540, 2> X3Narrow;   // 128 x 64, 4 waves of 64 x 32 (72 KB of LDS): two per CU

NtSwitches &nt_switches() {
    static NtSwitches sw = [] {
        NtSwitches s;
        const char *e = getenv("PDGN_GEMM");
        s.mode = (e && e[0] == 'f') ? 0 : (e && e[0] == 'x' && e[1] == '3') ? 1 : 2;      // fp32 | x3 | x2 (default: two parts where they pay)
        e = getenv("PDGN_NT_CFG");
        s.cfg = (e && *e) ? atoi(e) : -1;
        e = getenv("PDGN_X3_SPLITK");                 // 0: weight gradients on the flattened stream-K order (A/B arm)
        s.splitk = !(e && e[0] == '0');
        // the bf16 matrix instruction per instance class: bit (4 tile + class) of shape16 set = v_mfma_f32_16x16x32_bf16; tile 0 256x128,
        // 1 128x128, 2 128x64; class 0 plain, 1 pre-split second operand, 2 atomic (stream-K tails, weight gradients), 3 extended
        // epilogue.  PDGN_X3_SHAPE = 32 / 16: every class; PDGN_X3_SHAPE16_MASK = the mask itself (hex); default X3_DEFAULT_MASK
        s.shape16 = X3_DEFAULT_MASK;
        e = getenv("PDGN_X3_SHAPE");
        if (e && *e) s.shape16 = atoi(e) == 16 ? 0xfff : 0;
        e = getenv("PDGN_X3_SHAPE16_MASK");
        if (e && *e) s.shape16 = (int)strtol(e, nullptr, 16) & 0xfff;
        s.shape16_default = s.shape16;
        return s;
    }();
    return sw;
}

// mode: 1 = bf16 matrix cores (three bf16 parts, six partial products per fp32 product), 2 = fp16 matrix cores (two scaled fp16
// parts, three partial products), 0 = fp32 matrix instructions, < 0 = leave.  Returns the mode in force before the call.
extern "C" int pdgn_gemm_set_mode(int mode) {
    const int old = nt_switches().mode;
    if (mode >= 0) nt_switches().mode = mode > 2 ? 1 : mode;
    return old;
}

// Two-part mode: the arena of the library's own maxima scans (device memory that stays the library's to use until replaced; NULL / 0
// detaches; used round-robin in 1-KB units: an operand of r kernel-side rows takes 4 r bytes).  Without one the contractions of
// mode 2 are refused (-1) unless both operands' maxima are handed in.
extern "C" int pdgn_gemm_set_scale_slots(void *slots, long long bytes) {
    if (bytes < 0 || ((uintptr_t)slots & 15)) return PDGN_ERR_INVALID;
    const long long n = bytes / 1024;
    x2_ring_slots.store(0);
    x2_ring.store((unsigned *)slots);
    x2_ring_slots.store(slots ? (unsigned)(n > 0x7fffffff ? 0x7fffffff : n) : 0u);
    return 0;
}

// Two-part mode: the maxima of |x| (bit patterns) of every row (rowmax[rows]; may be NULL) and / or every column (colmax[cols]; may
// be NULL) of an fp32 matrix (rows x cols, pitch ld; cols and ld multiples of 4, src 16-byte aligned) in ONE pass on `stream`, and
// the hand-over of such arrays for the operands of the calling thread's next contraction call: max_a[i] / max_w[j] = the maximum
// of row i of the first / row j of the second operand AS THAT ENTRY POINT'S KERNEL SEES THEM -- pdgn_gemm_nt / _nt_ps: rows of A
// (m) and rows of W (n); pdgn_gemm_nn: rows of A (m) and COLUMNS of Wt (n); pdgn_gemm_tn_big: columns of dY (n) and columns of X
// (k).  NULL: scanned by the call.  Upper bounds are as good as maxima as long as they are finite (a bound 2^b above the row's
// maximum costs b of the 16 binades over which a value keeps its full 22 bits).
extern "C" int pdgn_absmax_rows_cols(long long rows, int cols, const float *src, int ld, unsigned *rowmax, unsigned *colmax,
                                     pdgn_stream_t stream) {
    if (!src) return PDGN_ERR_INVALID;
    return x2_maxima_launch(src, rows, cols, ld, rowmax, colmax, (hipStream_t)stream);
}
extern "C" int pdgn_gemm_set_operand_scales(const unsigned *max_a, const unsigned *max_w) {
    x2_next_max_a = max_a;
    x2_next_max_w = max_w;
    return 0;
}

// cfg: -1 = the launch model's pick, 0 .. 3 = force a tile configuration (measurement / tests), < -1 = leave.  Returns the
// previous value.
extern "C" int pdgn_gemm_set_config(int cfg) {
    const int old = nt_switches().cfg;
    if (cfg >= -1) nt_switches().cfg = cfg > 3 ? 3 : cfg;
    return old;
}

// shape: 32 = v_mfma_f32_32x32x16_bf16, 16 = v_mfma_f32_16x16x32_bf16 for EVERY later launch, -1 = back to the process default (the
// built-in per-class mask or what the environment said),
// 0x1000 | mask = that per-class mask (nt_switches), anything else = leave.  Returns the mask in force before the call (tests /
// tools: both shapes run through every direct GEMM test).
extern "C" int pdgn_gemm_set_shape(int shape) {
    const int old = nt_switches().shape16;
    if (shape == 16) nt_switches().shape16 = 0xfff;
    else if (shape == 32) nt_switches().shape16 = 0;
    else if (shape == -1) nt_switches().shape16 = nt_switches().shape16_default;
    else if (shape >= 0x1000 && shape <= 0x1fff) nt_switches().shape16 = shape & 0xfff;
    return old;
}

static int x3_mode() { return nt_switches().mode; }
static bool x3_shape16(int cfg_index, bool atomic, bool epi, bool pw) {
    const int cls = epi ? 3 : atomic ? 2 : pw ? 1 : 0;
    return (nt_switches().shape16 >> (4 * cfg_index + cls)) & 1;
}

static int x3_pick(long long m, int n, int k, bool stats) {
    const int forced = nt_switches().cfg;          // 0 .. 2 (3, the fp32 kernel's fourth, reads as 2)
    if (forced >= 0) return forced > 2 ? 2 : forced;
    const bool sk = !stats;
    const double c[3] = {X3Big::plan(m, n, k, sk).cost, X3Square::plan(m, n, k, sk).cost, X3Narrow::plan(m, n, k, sk).cost};
    int best = 0;
    for (int i = 1; i < 3; ++i)
        if (c[i] < c[best]) best = i;
    return best;
}

// Whether a product against PRE-SPLIT planes should use two-part ones (pdgn_split_f16x2) when scan_bytes of the activations would
// still have to be scanned for their maxima: with nothing to convert for the weight the eight-wave two-part loop on the 256 x 128
// tile is ahead of every three-part tile from ~2 GFLOP on (tools/cfg_probe.py: 17920 x 256 x 2560 153 -> 98 us, 35840 x 512 x 256
// 68 -> 47, 17920 x 6432 x 64 136 -> 107), as long as the scan stays under 4.5 bytes per kflop.
extern "C" int pdgn_gemm_two_part_planes(long long m, int n, int k, long long scan_bytes) {
    if (nt_switches().mode != 2 || m < 256 || n < 4 || k < 32) return 0;
    const double flops = 2.0 * (double)m * n * k;
    return (flops >= 2e9 && (double)scan_bytes <= 4.5e-3 * flops) ? 1 : 0;
}
// The tail workspace of pdgn_gemm_nt_ps(m, n, k) for planes of `parts` parts (two-part planes always run on the 256 x 128 tile).
extern "C" long long pdgn_gemm_nt_ps_workspace_floats(long long m, int n, int k, int parts, int with_stats);

// Whether the contraction (m, n, k) runs on two parts under the switches in force when scan_bytes of its operands still have to be
// scanned for their maxima (x2_pays): what a caller that pre-splits a weight (pdgn_split_f16x2 or _bf16x3) or hands maxima over asks first.
extern "C" int pdgn_gemm_two_part(long long m, int n, int k, long long scan_bytes) {
    if (!x3_mode() || m < 1 || n < 4 || k < 4) return 0;
    return x2_pays(x3_pick(m, n, k, false), m, n, k, scan_bytes) ? 1 : 0;
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
                       const NtEpi &epi = NtEpi(), const unsigned short *Wp = nullptr, long long wplane = 0, int parts = 3) {
    // two-part planes run on the 256 x 128 tile whatever the launch model would pick for three parts (with the maxima handed in
    // and nothing to convert for the weight the eight-wave two-part loop is ahead from ~2 GFLOP on: tools/cfg_probe.py); a
    // launch that emits BatchNorm partials keeps the geometry pdgn_gemm_nt_stat_rows / _stat_block_rows promised
    int cfg = x3_pick(m, n, k, stat_part != nullptr || epi.any());
    if (Wp && parts == 2) {
        if (stat_part && cfg != 0) return PDGN_ERR_INVALID;
        cfg = 0;
    }
    switch (cfg) {
        case 0: return X3Big::launch<WT>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, s, epi, false, Wp, wplane, parts);
        case 2: return X3Narrow::launch<WT>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, s, epi, false, Wp, wplane, parts);
        default: return X3Square::launch<WT>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, s, epi, false, Wp, wplane, parts);
    }
}

// C (m x n, row pitch ldc) = A (m x k, pitch lda) W (n x k, pitch ldw)^T (+ bias[n]) (+ addend (m x n, pitch ldadd)).
// n, k and every pitch are multiples of 4 floats and all base pointers 16-byte aligned.  stat_part (may be NULL):
// pdgn_gemm_nt_stat_rows(m, n, k) rows of [3n] floats = per-column sum (x - pv) | sum (x - pv)^2 | pv of row blocks of C
// (pv: the block's first row; pdgn_gemm_nt_stat_block_rows rows per block) for pdgn_bn_stats_from_gemm_partials.
extern "C" int pdgn_gemm_nt(long long m, int n, int k, const float *A, int lda, const float *W, int ldw,
                            const float *bias, const float *addend, int ldadd, float *C, int ldc, float *stat_part,
                            pdgn_stream_t stream) {
    const X3Handover handover;                                 // (pending workspace / maxima belong to this call, even if it is refused)
    if (!x3_mode()) return fp32_gemm_nt(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, stream);
    if (!x3_args_ok(m, n, k, lda, ldw, ldadd, ldc, addend, false)) return PDGN_ERR_INVALID;
    return x3_dispatch<false>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream);
}

// The same product with the second operand given transposed: C (m x n) = A (m x k) Wt (k x n, row pitch ldw) -- the
// input gradient dX = dY W of a dense layer straight from its (C_out x C_in) weight.
extern "C" int pdgn_gemm_nn(long long m, int n, int k, const float *A, int lda, const float *Wt, int ldw,
                            const float *bias, const float *addend, int ldadd, float *C, int ldc, float *stat_part,
                            pdgn_stream_t stream) {
    const X3Handover handover;                                 // (pending workspace / maxima belong to this call, even if it is refused)
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
    const X3Handover handover;                                 // (pending workspace / maxima belong to this call, even if it is refused)
    if (!x3_mode())
        return fp32_gemm_nt_ex(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, row_bias, ld_rb, rows_per_group,
                               act, gate, ldgate, transposed_w, stream);
    if (!x3_args_ok(m, n, k, lda, ldw, ldadd, ldc, addend, transposed_w != 0)) return PDGN_ERR_INVALID;
    if ((act != 0 && act != 2) || (row_bias && (ld_rb < n || ld_rb % 4 || rows_per_group < 1)) ||
        (gate && (ldgate < n || ldgate % 4)) || m >= (1LL << 31) || (row_bias && m * rows_per_group >= (1LL << 32)) ||
        (row_bias && rows_per_group >= 1 && ((m + rows_per_group - 1) / rows_per_group) * (long long)ld_rb * 4 >= (long long)NT_OOB))
        return PDGN_ERR_INVALID;
    NtEpi e;
    e.row_bias = row_bias; e.ld_rb = ld_rb; e.rows_per_group = rows_per_group; e.act = act; e.gate = gate; e.ldgate = ldgate;
    return transposed_w ? x3_dispatch<true>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream, e)
                        : x3_dispatch<false>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream, e);
}

// pdgn_gemm_nt / pdgn_gemm_nt_ex with the second operand PRE-SPLIT by pdgn_split_bf16x3: Wplanes = three bf16 planes [n][ldw]
// (wplane elements apart) of the (n x k) weight -- or of the transpose of a (k' x n') weight, which makes this the input-gradient
// product dX = dY W as well.  Bit-identical to the unsplit entry points on the bf16 matrix cores (the planes hold exactly the
// parts the loader would compute); only there: with pdgn_gemm_set_mode(0) in force the call is refused (-1).
extern "C" int pdgn_gemm_nt_ps(long long m, int n, int k, const float *A, int lda, const unsigned short *Wplanes, int ldw,
                               long long wplane, int parts, const float *bias, const float *addend, int ldadd, float *C, int ldc,
                               float *stat_part, const float *row_bias, int ld_rb, int rows_per_group, int act, const float *gate,
                               int ldgate, pdgn_stream_t stream) {
    const X3Handover handover;                                 // (pending workspace / maxima belong to this call, even if it is refused)
    if (!x3_mode() || !Wplanes || (parts != 3 && parts != 2)) return PDGN_ERR_INVALID;
    if (!x3_args_ok(m, n, k, lda, ldw, ldadd, ldc, addend, false)) return PDGN_ERR_INVALID;
    if (wplane < (long long)(n - 1) * ldw + k || wplane >= (1LL << 28) || ldw % 8 || wplane % 8 || ((uintptr_t)Wplanes & 15))
        return PDGN_ERR_INVALID;                                  // 16-B loads of 8 bf16: rows and planes start on 16-B boundaries
    if ((act != 0 && act != 2) || (row_bias && (ld_rb < n || ld_rb % 4 || rows_per_group < 1)) ||
        (gate && (ldgate < n || ldgate % 4)) || m >= (1LL << 31) || (row_bias && m * rows_per_group >= (1LL << 32)) ||
        (row_bias && rows_per_group >= 1 && ((m + rows_per_group - 1) / rows_per_group) * (long long)ld_rb * 4 >= (long long)NT_OOB))
        return PDGN_ERR_INVALID;
    if (parts == 2 && x3_mode() == 2 && !bias && !addend && !row_bias && !act && !gate && rp_takes(m, n, k)) {
        // a short reduction and a wide result: the row-panel kernel (gemm_rp.hip: A resident in registers, the weight tiles streamed,
        // the same two-part arithmetic bit for bit)
        // (A's row maxima: taken inside the kernel -- handed-in ones are not needed and not read)
        return rp_launch(m, n, k, A, lda, Wplanes, ldw, wplane, C, ldc, stat_part, (hipStream_t)stream);      // (stat_part: one partial row per 256-row panel, pdgn_gemm_nt_ps_stat_rows)
    }
    NtEpi e;
    e.row_bias = row_bias; e.ld_rb = ld_rb; e.rows_per_group = rows_per_group > 0 ? rows_per_group : 1; e.act = act; e.gate = gate;
    e.ldgate = ldgate;
    return x3_dispatch<false>(m, n, k, A, lda, nullptr, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream, e, Wplanes,
                              wplane, parts);
}

// Stream-K tails without atomics: floats of workspace the tail of pdgn_gemm_nt / _nn / _nt_ps (m, n, k) wants (0: that launch has no
// tail, or runs on the fp32 instructions), and the hand-over of a buffer to the NEXT such call of the calling thread.
extern "C" long long pdgn_gemm_tail_workspace_floats(long long m, int n, int k, int with_stats) {
    if (!x3_mode() || m < 1 || n < 4 || k < 4) return 0;
    const bool sk = !with_stats;
    switch (x3_pick(m, n, k, with_stats != 0)) {
        case 0: return X3Big::tail_floats(m, n, k, sk);
        case 2: return X3Narrow::tail_floats(m, n, k, sk);
        default: return X3Square::tail_floats(m, n, k, sk);
    }
}
extern "C" long long pdgn_gemm_nt_ps_workspace_floats(long long m, int n, int k, int parts, int with_stats) {
    if (!x3_mode() || m < 1 || n < 4 || k < 4) return 0;
    if (parts == 2 && rp_takes(m, n, k)) return 0;                        // (the row-panel kernel has no tail)
    if (parts == 2) return with_stats ? 0 : X3Big::tail_floats(m, n, k, true);
    return pdgn_gemm_tail_workspace_floats(m, n, k, with_stats);
}
extern "C" int pdgn_gemm_set_tail_workspace(float *ws, long long floats) {
    if (floats < 0 || ((uintptr_t)ws & 15)) return PDGN_ERR_INVALID;
    x3_tail_ws = ws;
    x3_tail_ws_floats = ws ? floats : 0;
    return 0;
}

// Which kernel instance and grid the plain (no bias / addend / epilogue) data-parallel launch of pdgn_gemm_nt_ps(m, n, k, ...) uses under
// the switches in force: *sym = the instance's host symbol (NULL for an instance that lives in another translation unit: the
// 16x16x32 arm), *grid = its grid.  Lets a measurement find that launch inside a recorded iteration (pdgn_replay_kernel_nodes).
extern "C" int pdgn_gemm_nt_ps_launch_info(long long m, int n, int k, int parts, const void **sym, int *grid) {
    if (!x3_mode() || m < 1 || n < 4 || k < 4 || !sym || !grid) return PDGN_ERR_INVALID;
    const int cfg = x3_pick(m, n, k, false);
    const bool s16 = x3_shape16(cfg, false, false, true);
    *sym = nullptr;
    if (parts == 2) {
        *grid = X3Big::plan(m, n, k, true).grid_dp;
        *sym = x3_symbol_h2(16, true);
    } else if (cfg == 0) {
        *grid = X3Big::plan(m, n, k, true).grid_dp;
        if (!s16) *sym = (const void *)gemm_x3_kernel<4, 2, 2, 2, 1, false, false, false, false, true, 32>;
    } else if (cfg == 2) {
        *grid = X3Narrow::plan(m, n, k, true).grid_dp;
        if (!s16) *sym = (const void *)gemm_x3_kernel<2, 1, 2, 2, 2, false, false, false, false, true, 32>;
    } else {
        *grid = X3Square::plan(m, n, k, true).grid_dp;
        if (!s16) *sym = (const void *)gemm_x3_kernel<2, 2, 2, 2, 1, false, false, false, false, true, 32>;
    }
    return 0;
}

// Host symbols of the two small kernels a contraction call may launch around its matrix-core kernels (the reduce of a stream-K
// tail's partial tiles; the scan of an operand's maxima): for measurements that take a whole call out of a recorded iteration.
extern "C" int pdgn_gemm_aux_symbols(const void **reduce, const void **scan) {
    if (reduce) *reduce = (const void *)x3_sk_reduce_kernel;
    if (scan) *scan = (const void *)x2_maxima_kernel<8>;       // (the instance a scan of conv2's first operand takes: 5120 columns in slabs of 2048)
    return 0;
}

// Weight gradient of a point-major dense layer, dW (n x k) = dY (m x n)^T X (m x k): the same kernel with BOTH operands
// given transposed, the reduction over the m rows split over the workgroups (stream-K launch, fp32 atomics into dW,
// which the launch zero-fills itself).  For outputs of at least one 128 x 64 tile; pdgn_gemm_tn (gemm_tn.hip) keeps the
// small ones.
extern "C" int pdgn_gemm_tn_big(long long m, int n, int k, const float *dY, int ldy, const float *X, int ldx, float *dW,
                                int dw_is_zero, pdgn_stream_t stream) {
    const X3Handover handover;                                 // (pending workspace / maxima belong to this call, even if it is refused)
    if (!x3_mode()) return fp32_gemm_tn_big(m, n, k, dY, ldy, X, ldx, dW, stream);       // (zero-fills dW itself in any case)
    if (m < 1 || n < 4 || k < 4 || n % 4 || k % 4 || ldy % 4 || ldx % 4 || ldy < n || ldx < k || ldy >= (1 << 19) ||
        ldx >= (1 << 19) || m > 0x7fffffffLL * 16)
        return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    // kernel roles: output rows = n (columns of dY), output columns = k (columns of X), reduction = m
    const int cfg = x3_pick(n, k, (int)(m > 0x7fffffff ? 0x7fffffff : m), false);
    switch (cfg) {
        case 0: return X3Big::launch<true, true>(n, k, (int)m, dY, ldy, X, ldx, nullptr, nullptr, 0, dW, k, nullptr, s, NtEpi(), dw_is_zero != 0);
        case 2: return X3Narrow::launch<true, true>(n, k, (int)m, dY, ldy, X, ldx, nullptr, nullptr, 0, dW, k, nullptr, s, NtEpi(), dw_is_zero != 0);
        default: return X3Square::launch<true, true>(n, k, (int)m, dY, ldy, X, ldx, nullptr, nullptr, 0, dW, k, nullptr, s, NtEpi(), dw_is_zero != 0);
    }
}

// Floats of workspace with which pdgn_gemm_tn_big(m, n, k) sums its k slices without atomics (pdgn_gemm_set_tail_workspace hands the
// buffer to the next call, as for the stream-K tails); 0: that call does not split (or runs on the fp32 instructions).
extern "C" long long pdgn_gemm_tn_big_workspace_floats(long long m, int n, int k) {
    if (!x3_mode() || m < 1 || n < 4 || k < 4) return 0;
    const int red = (int)(m > 0x7fffffff ? 0x7fffffff : m);
    switch (x3_pick(n, k, red, false)) {
        case 0: return X3Big::at_floats(n, k, red);
        case 2: return X3Narrow::at_floats(n, k, red);
        default: return X3Square::at_floats(n, k, red);
    }
}

// The same two questions for pdgn_gemm_nt_ps with `parts`-part planes, bias / addend / epilogue extras absent (`plain` != 0): a
// short-reduction product on two-part planes runs on the row-panel kernel (gemm_rp.hip), whose partial rows cover a 256-row panel each.
extern "C" int pdgn_gemm_nt_ps_row_panel(long long m, int n, int k, int parts, int plain) {
    return (x3_mode() == 2 && parts == 2 && plain && m >= 1 && rp_takes(m, n, k)) ? 1 : 0;
}
extern "C" long long pdgn_gemm_nt_ps_stat_rows(long long m, int n, int k, int parts, int plain) {
    if (x3_mode() == 2 && parts == 2 && plain && m >= 1 && rp_takes(m, n, k)) return (m + 255) / 256;
    return pdgn_gemm_nt_stat_rows(m, n, k);
}
extern "C" int pdgn_gemm_nt_ps_stat_block_rows(long long m, int n, int k, int parts, int plain) {
    if (x3_mode() == 2 && parts == 2 && plain && m >= 1 && rp_takes(m, n, k)) return 256;
    return pdgn_gemm_nt_stat_block_rows(m, n, k);
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
#endif  // X3_KERNEL_ONLY
