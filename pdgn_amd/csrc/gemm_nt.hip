// gemm_nt.hip -- the dense contractions of the point-major layers on the fp32 matrix cores.
//
//   C[m, n] = sum_k A[m, k] * W[n, k]  (+ bias[n]) (+ addend[m, n])        A (M x K), W (N x K) row-major
//
// i.e. rows x C_in @ (C_out x C_in)^T -- every conv / linear of the deconvolution stack and of the
// discriminators once activations are point-major (models/PDGNet_v2.py:559-625, 835-862, 886-1014), and,
// called with the transposed weight, their input gradients dX = dY W.  Optional epilogue: per-column
// partial sums / sums of squares of each row block, so the BatchNorm that follows needs no statistics pass.
//
// Structure (gfx950, wave64):
//   * PERSISTENT workgroups: the grid is one or two workgroups per CU; each walks a sequence of work items
//     (tile, k range) and streams ONE continuous sequence of 32-deep k chunks through a ring of LDS stages --
//     the loads of the next item's first chunks are in flight while the current item finishes, so short
//     reductions (K = 128: four chunks per tile) pay no pipeline fill per tile;
//   * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 B per lane, no staging
//     registers, out-of-range rows / k columns read as zeros through the buffer descriptor); a chunk is
//     [rows][32 floats] with the 16-B column of row r stored at position (col ^ (r & 7)) -- applied on the
//     SOURCE address, the LDS image itself is lane-linear -- which makes every ds_read_b128 of the fragment
//     reads bank-conflict-free;
//   * one barrier per chunk; waits on the DMA are COUNTED (s_waitcnt vmcnt(n)): STAGES-1 chunks stay in
//     flight across barriers and across the epilogue's stores;
//   * v_mfma_f32_16x16x4_f32 (exact fp32 fma chains): a lane feeds four consecutive MFMA k-steps from ONE
//     ds_read_b128 per operand row (lane (i, g) holds k = 16q + 4g + u for step u; A and W agree, and the
//     MFMA sums over k, so the assignment of k values to (step, lane group) is free).  The weight rows are
//     the MFMA's row operand, so a lane ends up with FOUR CONSECUTIVE output columns: one 16-B store each;
//   * work decomposition by the host: whole tiles dealt round-robin over the workgroups ("data-parallel"
//     launch, XCD-aware order: the workgroups of one XCD hold neighbouring tiles, which share operand panels
//     in that XCD's L2), and, when the tile count does not fill the last round of CUs, the remaining tiles
//     in a second launch that splits the flattened (tile, k chunk) space evenly ("stream-K" launch, partial
//     tiles added with fp32 atomics into the zero-filled rows).
#include "gemm_shared.h"

template <int TM, int TN, int WM, int WN, bool ATOMIC, bool WT, bool AT, bool EPI = false>   // EPI: the extended epilogue of pdgn_gemm_nt_ex
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_nt_kernel(const NtArgs p) {
    constexpr int NW = WM * WN, BM = 16 * TM * WM, BN = 16 * TN * WN;
    // WT: the weight operand is given TRANSPOSED, W^T (K x N) row-major (the input gradient dX = dY W uses the layer's own
    // (C_out x C_in) weight as it is): its chunk is [32 k][BN] in LDS, 1-KB pieces padded apart so that the b32 fragment
    // reads of rows k and k + 4 fall on different banks
    // AT: likewise the FIRST operand, A^T (K x M) row-major: with both, C = At^T Wt is the weight gradient dW = dY^T X of a
    // dense layer, the reduction running over the (10^4 .. 10^5.5) rows and split over the workgroups by the stream-K launch.
    constexpr int WROWS = 256 / BN, WPAD = WT ? 4 * WROWS : 0;      // k rows per 1-KB piece, pad floats after each piece
    constexpr int AROWS = 256 / BM, APAD = AT ? 4 * AROWS : 0;
    static_assert(!AT || (BM <= 256 && 256 % BM == 0 && 4 % AROWS == 0), "A^T pieces: whole rows, a divisor of 4 per piece");
    static_assert(!WT || (BN <= 256 && 256 % BN == 0 && 4 % WROWS == 0), "W^T pieces: whole rows, a divisor of 4 per piece");
    constexpr int A_FLOATS = BM * NT_BK + (BM / 8) * APAD;          // A part of a stage
    constexpr int STAGE_FLOATS = A_FLOATS + BN * NT_BK + (BN / 8) * WPAD;
    constexpr int NPA = BM / 8 / NW, NPB = BN / 8 / NW, NP = NPA + NPB;   // 1-KB DMA pieces per wave and chunk
    constexpr int NS = ATOMIC ? 0 : TM * TN;                              // counted stores per wave and item
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "pieces must divide over the waves");
    constexpr int STAGES = 2;
    static_assert(4 * TM * TN >= 2 * NP, "a half chunk must have room for the DMA pieces between its MFMAs");
    __shared__ __attribute__((aligned(1024))) float smem[STAGES * STAGE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;

    // XCD-aware decode of the 1-D grid: ids are dealt round-robin to the 8 XCDs; XCD x takes a contiguous range
    const int G = gridDim.x, pid = blockIdx.x;
    const int gq = G >> 3, gr = G & 7, xcd = pid & 7;
    const int v = xcd * gq + min(xcd, gr) + (pid >> 3);

    const int KC = p.kchunks;
    // Tile index -> (tile row, tile column), GROUPED: NT_GROUP_M tile rows form a group that is walked column by column
    // (tile row fastest), so the ~64 tiles an XCD works on at a time are an 8 x 8 block sharing 8 activation and 8
    // weight panels through its L2 -- in plain row-major order they would be one activation panel and 64 weight
    // panels, each weight panel read once per tile row by every XCD (measured on the per-point GEMM: 1.87 GB of
    // operand re-reads from beyond L2 for 25 MB of operands).
    auto decode = [&](int tile, int &tm, int &tn) {
        const int per_group = NT_GROUP_M * p.tiles_n;
        const int grp = tile / per_group, r = tile - grp * per_group;
        const int first = grp * NT_GROUP_M;
        const int gsz = min(NT_GROUP_M, p.tiles_m - first);
        tn = r / gsz;
        tm = first + (r - tn * gsz);
    };
    // ---- work-item cursors (all scalar).  DP: item j = tile tile_begin + v + j*G.  SK: a flattened range.
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

    // descriptors of the load cursor's tile (a transposed operand: of its current chunk -- the rows of a chunk start
    // k0 * pitch floats into the matrix, which a 32-bit offset from the tile's origin could not always reach)
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
    // One DMA piece of the load cursor's chunk (i < NPA: activation rows, else weight rows) into `stage`.
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

    // ---- fragment read offsets (floats): row (li) * 32 + 4 * ((4q + lg) ^ (li & 7)), q = 0, 1
    const int fo0 = li * NT_BK + 4 * ((lg) ^ (li & 7));
    const int fo1 = li * NT_BK + 4 * ((4 + lg) ^ (li & 7));
    const int abase = wm * 16 * TM * NT_BK, bbase = A_FLOATS + wn * 16 * TN * NT_BK;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 ra0[TM], rb0[TN], ra1[TM], rb1[TN];                     // fragments of the two 16-deep halves of a chunk
    // WT: k row r of the W^T chunk sits at float (r / WROWS) * (256 + WPAD) + (r % WROWS) * BN; a lane's fragment element u
    // of column block b is row 16q + 4*lg + u, column wn*16*TN + 16b + li
    const int wt_lane = ((4 * lg) / WROWS) * (256 + WPAD) + ((4 * lg) % WROWS) * BN + wn * 16 * TN + li;
    const int at_lane = ((4 * lg) / AROWS) * (256 + APAD) + ((4 * lg) % AROWS) * BM + wm * 16 * TM + li;
    auto read_half = [&](int stage, int q, float4 *ra, float4 *rb) {
        if (!AT) {
            const float *sa = smem + stage * STAGE_FLOATS + abase + (q ? fo1 : fo0);
#pragma unroll
            for (int a = 0; a < TM; ++a) ra[a] = *reinterpret_cast<const float4 *>(sa + a * 16 * NT_BK);
        } else {
            const float *sa = smem + stage * STAGE_FLOATS + at_lane;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 16 * q + u;
                    e[u] = sa[(r / AROWS) * (256 + APAD) + (r % AROWS) * BM + 16 * a];
                }
                ra[a] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        if (!WT) {
            const float *sb = smem + stage * STAGE_FLOATS + bbase + (q ? fo1 : fo0);
#pragma unroll
            for (int b = 0; b < TN; ++b) rb[b] = *reinterpret_cast<const float4 *>(sb + b * 16 * NT_BK);
        } else {
            const float *sb = smem + stage * STAGE_FLOATS + A_FLOATS + wt_lane;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 16 * q + u;                      // + 4*lg, folded into wt_lane (4*lg and r never carry across a piece: WROWS | 4)
                    e[u] = sb[(r / WROWS) * (256 + WPAD) + (r % WROWS) * BN + 16 * b];
                }
                rb[b] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };
    // ---- output of a finished item.  Direct mode: D row (4*lg + r) = weight row, D column li = activation row, so a lane
    // holds 4 consecutive output columns of row li; the 16-B stores of item t are issued from INSIDE the first half chunk
    // of item t+1 (store of accumulator (a, b), then the MFMA that restarts it from zero), where their issue time falls
    // into the shadow of running MFMAs and no register is copied or zero-filled.
    const int mloc0 = wm * 16 * TM + li, nloc0 = wn * 16 * TN + 4 * lg;
    __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)p.C, 0, 0, 0x00020000);   // nothing pending: all out of range
    unsigned st_off[TN];                                           // byte offset of (row mloc0, column block b), or out of range
#pragma unroll
    for (int b = 0; b < TN; ++b) st_off[b] = NT_OOB;
    auto store_acc = [&](int a, int b) {
        const unsigned off = st_off[b] + (unsigned)(a * 16 * p.ldc) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[a][b]), rsC, off, 0, 0);
    };
    // bias / addend / statistics on the finished accumulators (in place), descriptor + offsets of its stores
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
            for (int b = 0; b < TN; ++b)
                st_off[b] = (nloc0 + 16 * b < ncols && !(p.dbg & 1)) ? (unsigned)(mloc0 * p.ldc + nloc0 + 16 * b) * 4u : NT_OOB;
            if (p.addend) {                                        // all tile loads in flight at once; out of range reads 0
                __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)(p.addend + m0 * p.ldadd + n0), 0,
                                                                               (int)(mrows * p.ldadd * 4), 0x00020000);
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const int ml = mloc0 + 16 * a, nl = nloc0 + 16 * b;
                        acc[a][b] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                 rsD, nl < ncols ? (unsigned)(ml * p.ldadd + nl) * 4u : NT_OOB, 0, 0));
                    }
            }
            if (p.bias) {
                __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)(p.bias + n0), 0, ncols * 4, 0x00020000);
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const f32x4 bz = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(nloc0 + 16 * b) * 4u, 0, 0));
#pragma unroll
                    for (int a = 0; a < TM; ++a) acc[a][b] += bz;
                }
            }
            if (EPI && p.row_bias) {                               // a bias per group of rows: the heads' per-sample term
                // (bounded by the table's own extent -- groups x ld_rb floats, < 2^30 bytes by the entry point's check -- so that the
                // masked lanes' NT_OOB offset is out of range: an unbounded descriptor made them read row_bias + 1 GiB)
                __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)p.row_bias, 0, p.rb_bytes, 0x00020000);
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const unsigned row = (unsigned)(m0 + mloc0 + 16 * a);
                    const unsigned grp = row < (unsigned)p.M ? (p.rows_per_group == 1 ? row : __umulhi(row, p.rpg_magic)) : 0u;
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const int nl = nloc0 + 16 * b;
                        acc[a][b] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                 rsR, nl < ncols ? (grp * (unsigned)p.ld_rb + n0 + nl) * 4u : NT_OOB, 0, 0));
                    }
                }
            }
            if (EPI && p.act == 2) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[a][b][r] = acc[a][b][r] > 0.f ? acc[a][b][r] : 0.01f * acc[a][b][r];
            }
            if (EPI && p.gate) {                                   // all tile loads in flight at once; out of range reads 0
                __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void *)(p.gate + m0 * p.ldgate + n0), 0,
                                                                               (int)(mrows * p.ldgate * 4), 0x00020000);
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const int ml = mloc0 + 16 * a, nl = nloc0 + 16 * b;
                        const f32x4 g = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                     rsG, nl < ncols ? (unsigned)(ml * p.ldgate + nl) * 4u : NT_OOB, 0, 0));
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[a][b][r] *= g[r] > 0.f ? 1.f : 0.01f;
                    }
            }
            if (p.stat_part) {
                // per-column statistics of the wave's 16*TM rows, SHIFTED by the block's first row (pv): sum (x - pv),
                // sum (x - pv)^2 and pv itself -- E[x^2] - mean^2 on raw fp32 sums cancels quadratically in |mean| / std,
                // on sums shifted by a value of the same block it does not, and cl_finalize_blocks_kernel combines the
                // blocks in fp64 (Chan et al.).  In registers over a, then over the 16 lanes of equal lg; one partial
                // row of [3N] floats per (tile row, wave row).
                float *P = p.stat_part + ((size_t)tm * WM + wm) * 3 * p.N;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f}, pv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = __shfl(acc[0][b][r], lane & 48, 64);   // row li = 0 of row block a = 0
#pragma unroll
                    for (int a = 0; a < TM; ++a)
                        if (mloc0 + 16 * a < mrows) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float d = acc[a][b][r] - pv[r];
                                cs[r] += d;
                                cq[r] = __fmaf_rn(d, d, cq[r]);
                            }
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int d = 1; d < 16; d <<= 1) {
                            cs[r] += __shfl_xor(cs[r], d, 64);
                            cq[r] += __shfl_xor(cq[r], d, 64);
                        }
                    const int nl = nloc0 + 16 * b;
                    if (li == 0 && nl < ncols) {
                        *reinterpret_cast<float4 *>(P + n0 + nl) = make_float4(cs[0], cs[1], cs[2], cs[3]);
                        *reinterpret_cast<float4 *>(P + p.N + n0 + nl) = make_float4(cq[0], cq[1], cq[2], cq[3]);
                        *reinterpret_cast<float4 *>(P + 2 * p.N + n0 + nl) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                    }
                }
            }
        } else {
            // partial tile: D row (4*lg + r) = activation row, D column li = weight row (a register holds 16 consecutive
            // columns of 4 rows: 64-B atomic segments).  The holder of the tile's first chunk adds the bias / addend.
            const bool head = cp.kb == 0;
            const int mla = wm * 16 * TM + 4 * lg, nla = wn * 16 * TN + li;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int nl = nla + 16 * b;
                const bool nok = nl < ncols;
                const float bz = (head && p.bias && nok) ? p.bias[n0 + nl] : 0.f;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = mla + 16 * a + r;
                        if (nok && ml < mrows) {
                            float o = acc[a][b][r] + bz;
                            if (head && p.addend) o += p.addend[(m0 + ml) * p.ldadd + n0 + nl];
                            atomicAdd(p.C + (m0 + ml) * p.ldc + n0 + nl, o);
                        }
                    }
            }
        }
    };

    // The MFMAs of one 16-deep half chunk.
    //   MODE 0: plain.
    //   MODE 1: the NP pieces of the next refill are issued between them, one every (4 TM TN / NP) MFMAs: an LDS-DMA
    //           instruction occupies the wave's issue for tens of cycles, which must fall into the shadow of a running
    //           MFMA, not in front of the first one.
    //   MODE 2: first half chunk of an item: every accumulator is stored (the previous item's result; out of range when
    //           there is none) right before the MFMA that restarts it from zero.
    auto mfma_half = [&](const float4 *ra, const float4 *rb, auto mode) {
        constexpr int MODE = decltype(mode)::value;
        constexpr int TOTAL = 4 * TM * TN, EVERY = TOTAL / NP;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const float x = u == 0 ? ra[a].x : u == 1 ? ra[a].y : u == 2 ? ra[a].z : ra[a].w;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const float w = u == 0 ? rb[b].x : u == 1 ? rb[b].y : u == 2 ? rb[b].z : rb[b].w;
                    f32x4 c = acc[a][b];
                    if (MODE == 2 && u == 0) {
                        if (!ATOMIC) {
                            __builtin_amdgcn_sched_barrier(0);
                            store_acc(a, b);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        c = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                    acc[a][b] = ATOMIC ? __builtin_amdgcn_mfma_f32_16x16x4f32(x, w, c, 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, c, 0, 0, 0);
                    const int idx = (u * TM + a) * TN + b;
                    if (MODE == 1 && idx % EVERY == EVERY / 2 && idx / EVERY < NP) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(idx / EVERY);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    };
    typedef std::integral_constant<int, 0> M0_t;
    typedef std::integral_constant<int, 1> M1_t;
    typedef std::integral_constant<int, 2> M2_t;

    // ---- prologue: chunks 0 and 1 in flight, chunk 0 landed, its first half in registers
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
    read_half(0, 0, ra0, rb0);
    int stage = 0;
    while (cp.valid) {
        bool first = true;
        for (;;) {
            // (A) second half's fragments on their way while the first half is multiplied
            read_half(stage, 1, ra1, rb1);
            __builtin_amdgcn_sched_barrier(0);
            if (first) mfma_half(ra0, rb0, M2_t());
            else mfma_half(ra0, rb0, M0_t());
            // (B) everyone has read this stage to the end, and the next chunk has landed everywhere: my own pieces
            // (counted wait: only the stores of (A) were issued after them), then the barrier
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 63 ? 63 : NS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // (C) next chunk's first half on its way, this stage refilled with the chunk after next, while the second
            // half is multiplied
            first = false;
            todo--;
            if (todo > 0) read_half(stage ^ 1, 0, ra0, rb0);
            if (ld.valid) {
                issue_begin(stage);
                mfma_half(ra1, rb1, M1_t());
                advance_load();
            } else {
                mfma_half(ra1, rb1, M0_t());
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
            for (int b = 0; b < TN; ++b) store_acc(a, b);
    }
}

template <int TM, int TN, int WM, int WN, int MAXWG>
struct NtCfg {
    static constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
    static constexpr int LDS = 2 * (BM + BN) * NT_BK * 4;
    static constexpr int WG_PER_CU = (160 * 1024 / LDS) < MAXWG ? (160 * 1024 / LDS) : MAXWG;   // MAXWG: the register file's limit

    // what the launch will do: tiles, slots, whether a stream-K tail follows the whole rounds
    struct Plan {
        int tiles_m, tiles_n, kchunks, grid_dp, dp_tiles, grid_sk;
        long long sk_per_wg;
        double cost;                                               // launch model, microseconds
    };
    // Launch model (calibrated on MI355X, tools/gemm_shapes.py): a CU multiplies ~0.56 Mflop/us on this loop; the w
    // workgroups sharing a CU share that; a tile costs ~2.5 chunks on top of its own (ring fill, output); the atomics of
    // a partial tile are added at ~1.3 TB/s chip-wide.
    // Workgroups of the data-parallel launch: at least one per slot, and more -- each then walks fewer tiles -- so that
    // the hardware hands them out as CUs come free (this kernel shares the chip with the step's other streams; a static
    // assignment over exactly `slots` workgroups lets one delayed workgroup hold up the launch): enough tiles per
    // workgroup to keep ~32 chunks of work behind one ring fill.
    static int dp_grid(long long tiles, int slots, int kc) {
        static int tpw_env = -1;
        if (tpw_env < 0) { const char *e = getenv("PDGN_NT_TPW"); tpw_env = e ? atoi(e) : 0; }
        int tpw = tpw_env > 0 ? tpw_env : (32 + kc - 1) / kc;
        tpw = tpw < 1 ? 1 : (tpw > 8 ? 8 : tpw);
        if (tpw_env >= 1000) return (int)(tiles < slots ? tiles : slots);          // measurement: one workgroup per slot
        const long long g = (tiles + tpw - 1) / tpw;
        const long long lo = tiles < slots ? tiles : slots;
        return (int)(g < lo ? lo : g);
    }
    static double chunk_us(int w) { return 2.0 * BM * BN * NT_BK * w / (0.56e6 * (TM * TN >= 16 ? 1.0 : 0.95)); }
    static Plan plan(long long m, int n, int k, bool allow_sk) {
        Plan pl;
        pl.tiles_m = cdiv(m, BM);
        pl.tiles_n = cdiv(n, BN);
        pl.kchunks = cdiv(k, NT_BK);
        const long long T = (long long)pl.tiles_m * pl.tiles_n;
        const int cus = nt_cus(), slots = cus * WG_PER_CU, KC = pl.kchunks;
        const long long rounds = T / slots, tail = T - rounds * slots;
        const double ov = 2.5;
        const double full = (double)rounds * (KC + ov) * chunk_us(WG_PER_CU);
        // data-parallel only: the last, partial round runs ceil(tail / cus) workgroups on its busiest CU
        pl.dp_tiles = (int)T;
        pl.grid_dp = dp_grid(T, slots, KC);
        pl.grid_sk = 0;
        pl.sk_per_wg = 0;
        pl.cost = full + (tail ? (KC + ov) * chunk_us((int)((tail + cus - 1) / cus)) : 0.0);
        if (allow_sk && tail > 0) {
            // stream-K tail over g workgroups: every slot, one per CU, or fewer with at least 16 chunks each
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

    template <bool WT, bool AT = false>
    static int launch(long long m, int n, int k, const float *A, int lda, const float *W, int ldw, const float *bias,
                      const float *addend, int ldadd, float *C, int ldc, float *stat_part, hipStream_t s,
                      const NtEpi &epi = NtEpi()) {
        const bool allow_sk = stat_part == nullptr && ldc == n && !epi.any();
        const Plan pl = plan(m, n, k, allow_sk);
        NtArgs a;
        a.Wp = nullptr; a.wplane = 0;
        a.M = m; a.N = n; a.K = k; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldadd = ldadd;
        a.A = A; a.W = W; a.bias = bias; a.addend = addend; a.C = C; a.stat_part = stat_part;
        a.row_bias = epi.row_bias; a.ld_rb = epi.ld_rb; a.rows_per_group = epi.rows_per_group > 0 ? epi.rows_per_group : 1;
        a.rpg_magic = a.rows_per_group > 1 ? (unsigned)(0x100000000ULL / (unsigned)a.rows_per_group) + 1u : 0u;
        a.rb_bytes = epi.row_bias ? (int)((((m + a.rows_per_group - 1) / a.rows_per_group - 1) * (long long)epi.ld_rb + n) * 4) : 0;
        a.act = epi.act; a.gate = epi.gate; a.ldgate = epi.ldgate; a.sk_split = 0; a.sk_ws = nullptr; a.sk_seg = 0; a.max_a = a.max_w = nullptr;
        a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n; a.kchunks = pl.kchunks;
        a.dbg = 0;
#ifdef PDGN_NT_DEBUG
        { const char *e = getenv("PDGN_NT_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
        const long long T = (long long)pl.tiles_m * pl.tiles_n;
        if (pl.grid_sk) {
            // the tail tiles start in tile-row group dp_tiles / (NT_GROUP_M tiles_n): zero its rows and all below (whole rows; the data-parallel
            // launch overwrites its share of that tile row afterwards)
            const long long r0 = (long long)(pl.dp_tiles / (NT_GROUP_M * pl.tiles_n)) * NT_GROUP_M * BM;
            if (hipMemsetAsync(C + r0 * ldc, 0, (size_t)(m - r0) * ldc * sizeof(float), s) != hipSuccess) return pdgn_launch_status();
        }
        if (pl.grid_dp) {
            a.tile_begin = 0; a.tile_end = pl.dp_tiles; a.sk_per_wg = 0;
            if (!AT && epi.any())
                hipLaunchKernelGGL((gemm_nt_kernel<TM, TN, WM, WN, false, WT, false, true>), dim3(pl.grid_dp), dim3(64 * WM * WN), 0, s, a);
            else
                hipLaunchKernelGGL((gemm_nt_kernel<TM, TN, WM, WN, false, WT, AT>), dim3(pl.grid_dp), dim3(64 * WM * WN), 0, s, a);
        }
        if (pl.grid_sk) {
            a.tile_begin = pl.dp_tiles; a.tile_end = (int)T; a.sk_per_wg = pl.sk_per_wg;
            hipLaunchKernelGGL((gemm_nt_kernel<TM, TN, WM, WN, true, WT, AT>), dim3(pl.grid_sk), dim3(64 * WM * WN), 0, s, a);
        }
        return pdgn_launch_status();
    }
};

typedef NtCfg<4, 4, 4, 2, 1> NtBig;      // 256 x 128, 8 waves (96 KB): one workgroup per CU
typedef NtCfg<4, 4, 2, 2, 2> NtSquare;   // 128 x 128, 4 waves (64 KB): two per CU
typedef NtCfg<5, 2, 2, 2, 2> NtTall;     // 160 x 64, 4 waves (56 KB): two per CU; 35840 = 224 x 160
typedef NtCfg<4, 2, 2, 2, 3> NtNarrow;   // 128 x 64, 4 waves (48 KB, < 168 registers): three per CU

// Tile configuration for a problem: PDGN_NT_CFG (0-3, measurement only), else the cheapest by the launch model.
static int nt_pick(long long m, int n, int k, bool stats, bool no_tall = false) {
    const int forced = nt_switches().cfg;
    if (forced >= 0) return (no_tall && forced == 2) ? 1 : forced;
    const bool sk = !stats;
    const double c[4] = {NtBig::plan(m, n, k, sk).cost, NtSquare::plan(m, n, k, sk).cost, NtTall::plan(m, n, k, sk).cost,
                         NtNarrow::plan(m, n, k, sk).cost};
    int best = 1;
    for (int i = 0; i < 4; ++i)
        if (c[i] < c[best] * 0.97 && !(no_tall && i == 2)) best = i;    // the square tile unless another is clearly cheaper
    return best;
}

static bool nt_args_ok(long long m, int n, int k, int lda, int ldw, int ldadd, int ldc, const float *addend, bool wt) {
    return m >= 1 && n >= 4 && k >= 4 && n % 4 == 0 && k % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && ldc % 4 == 0 &&
           lda >= k && ldw >= (wt ? n : k) && ldc >= n && (!addend || (ldadd % 4 == 0 && ldadd >= n)) && lda < (1 << 19) &&
           ldw < (1 << 19) && ldc < (1 << 19) && (long long)cdiv(m, 128) * cdiv(n, 64) < 0x7fffffffLL &&
           (!wt || (long long)k * ldw < (1LL << 27));
}

template <bool WT>
static int nt_dispatch(long long m, int n, int k, const float *A, int lda, const float *W, int ldw, const float *bias,
                       const float *addend, int ldadd, float *C, int ldc, float *stat_part, hipStream_t s,
                       const NtEpi &epi = NtEpi()) {
    switch (nt_pick(m, n, k, stat_part != nullptr || epi.any())) {
        case 0: return NtBig::launch<WT>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, s, epi);
        case 2: return NtTall::launch<WT>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, s, epi);
        case 3: return NtNarrow::launch<WT>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, s, epi);
        default: return NtSquare::launch<WT>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, s, epi);
    }
}

// C (m x n, row pitch ldc) = A (m x k, pitch lda) W (n x k, pitch ldw)^T (+ bias[n]) (+ addend (m x n, pitch ldadd)).
// n, k and every pitch are multiples of 4 floats and all base pointers 16-byte aligned.  stat_part (may be NULL):
// pdgn_gemm_nt_stat_rows(m, n, k) rows of [3n] floats = per-column sum (x - pv) | sum (x - pv)^2 | pv of row blocks of C
// (pv: the block's first row; pdgn_gemm_nt_stat_block_rows rows per block) for pdgn_bn_stats_from_gemm_partials.
int fp32_gemm_nt(long long m, int n, int k, const float *A, int lda, const float *W, int ldw,
                            const float *bias, const float *addend, int ldadd, float *C, int ldc, float *stat_part,
                            pdgn_stream_t stream) {
    if (!nt_args_ok(m, n, k, lda, ldw, ldadd, ldc, addend, false)) return PDGN_ERR_INVALID;
    return nt_dispatch<false>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream);
}

// The same product with the second operand given transposed: C (m x n) = A (m x k) Wt (k x n, row pitch ldw) -- the
// input gradient dX = dY W of a dense layer straight from its (C_out x C_in) weight.
int fp32_gemm_nn(long long m, int n, int k, const float *A, int lda, const float *Wt, int ldw,
                            const float *bias, const float *addend, int ldadd, float *C, int ldc, float *stat_part,
                            pdgn_stream_t stream) {
    if (!nt_args_ok(m, n, k, lda, ldw, ldadd, ldc, addend, true)) return PDGN_ERR_INVALID;
    return nt_dispatch<true>(m, n, k, A, lda, Wt, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream);
}

// pdgn_gemm_nt / pdgn_gemm_nn (transposed_w != 0) with the extended epilogue:  C = act(A W^T + bias + addend +
// row_bias[row / rows_per_group]) * lrelu'(gate)  -- a bias per group of rows (the per-sample term of the generator's heads,
// models/PDGNet_v2.py:835-862 on cat([g broadcast, x])), LeakyReLU(0.01) on the result (act = 2), and / or the LeakyReLU
// derivative of a saved activation as a factor (gate: the result is the gradient wrt that layer's PRE-activation).  Each of
// them replaces a full elementwise pass over C.  No stream-K tail (whole tiles only), like a launch with statistics.
int fp32_gemm_nt_ex(long long m, int n, int k, const float *A, int lda, const float *W, int ldw, const float *bias,
                               const float *addend, int ldadd, float *C, int ldc, float *stat_part, const float *row_bias,
                               int ld_rb, int rows_per_group, int act, const float *gate, int ldgate, int transposed_w,
                               pdgn_stream_t stream) {
    if (!nt_args_ok(m, n, k, lda, ldw, ldadd, ldc, addend, transposed_w != 0)) return PDGN_ERR_INVALID;
    if ((act != 0 && act != 2) || (row_bias && (ld_rb < n || ld_rb % 4 || rows_per_group < 1)) ||
        (gate && (ldgate < n || ldgate % 4)) || m >= (1LL << 31) || (row_bias && m * rows_per_group >= (1LL << 32)) ||
        (row_bias && rows_per_group >= 1 && ((m + rows_per_group - 1) / rows_per_group) * (long long)ld_rb * 4 >= (long long)NT_OOB))
        return PDGN_ERR_INVALID;
    NtEpi e;
    e.row_bias = row_bias; e.ld_rb = ld_rb; e.rows_per_group = rows_per_group; e.act = act; e.gate = gate; e.ldgate = ldgate;
    return transposed_w ? nt_dispatch<true>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream, e)
                        : nt_dispatch<false>(m, n, k, A, lda, W, ldw, bias, addend, ldadd, C, ldc, stat_part, (hipStream_t)stream, e);
}

// Weight gradient of a point-major dense layer, dW (n x k) = dY (m x n)^T X (m x k): the same kernel with BOTH operands
// given transposed, the reduction over the m rows split over the workgroups (stream-K launch, fp32 atomics into dW,
// which the launch zero-fills itself).  For outputs of at least one 128 x 64 tile; pdgn_gemm_tn (gemm_tn.hip) keeps the
// small ones.
int fp32_gemm_tn_big(long long m, int n, int k, const float *dY, int ldy, const float *X, int ldx, float *dW,
                                pdgn_stream_t stream) {
    if (m < 1 || n < 4 || k < 4 || n % 4 || k % 4 || ldy % 4 || ldx % 4 || ldy < n || ldx < k || ldy >= (1 << 19) ||
        ldx >= (1 << 19) || m > 0x7fffffffLL * 16)
        return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    // kernel roles: output rows = n (columns of dY), output columns = k (columns of X), reduction = m
    switch (nt_pick(n, k, (int)(m > 0x7fffffff ? 0x7fffffff : m), false, true)) {
        case 0: return NtBig::launch<true, true>(n, k, (int)m, dY, ldy, X, ldx, nullptr, nullptr, 0, dW, k, nullptr, s);
        case 3: return NtNarrow::launch<true, true>(n, k, (int)m, dY, ldy, X, ldx, nullptr, nullptr, 0, dW, k, nullptr, s);
        default: return NtSquare::launch<true, true>(n, k, (int)m, dY, ldy, X, ldx, nullptr, nullptr, 0, dW, k, nullptr, s);
    }
}

// Number of [3n] partial-statistics rows pdgn_gemm_nt writes for this problem (tile rows x waves along m).
long long fp32_gemm_nt_stat_rows(long long m, int n, int k) {
    if (m < 1 || n < 1 || k < 1) return PDGN_ERR_INVALID;
    switch (nt_pick(m, n, k, true)) {
        case 0: return (long long)cdiv(m, NtBig::BM) * 4;
        case 2: return (long long)cdiv(m, NtTall::BM) * 2;
        case 3: return (long long)cdiv(m, NtNarrow::BM) * 2;
        default: return (long long)cdiv(m, NtSquare::BM) * 2;
    }
}

// Rows of C each of those partial rows covers (partial p: rows p * block .. ; the blocks past m are empty).
int fp32_gemm_nt_stat_block_rows(long long m, int n, int k) {
    if (m < 1 || n < 1 || k < 1) return PDGN_ERR_INVALID;
    switch (nt_pick(m, n, k, true)) {
        case 0: return NtBig::BM / 4;
        case 2: return NtTall::BM / 2;
        case 3: return NtNarrow::BM / 2;
        default: return NtSquare::BM / 2;
    }
}

// Tile configuration pdgn_gemm_nt picks for a problem (0: 256x128, 1: 128x128, 2: 160x64, 3: 128x64), + 16 when a
// stream-K launch follows the data-parallel one; host-side only.
int fp32_gemm_nt_config(long long m, int n, int k, int with_stats) {
    if (m < 1 || n < 1 || k < 1) return PDGN_ERR_INVALID;
    const int c = nt_pick(m, n, k, with_stats != 0);
    const bool sk = !with_stats;
    const int g = c == 0 ? NtBig::plan(m, n, k, sk).grid_sk : c == 1 ? NtSquare::plan(m, n, k, sk).grid_sk
                : c == 2 ? NtTall::plan(m, n, k, sk).grid_sk : NtNarrow::plan(m, n, k, sk).grid_sk;
    return c + (g ? 16 : 0);
}
