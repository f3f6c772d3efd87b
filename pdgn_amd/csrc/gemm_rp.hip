// gemm_rp.hip -- the SHORT-REDUCTION contractions of the point-major layers: C (M x N) = A (M x K) W^T with K = 32, 64 or 128 and a
// wide N -- the per-point tap product of a deconvolution block (models/PDGNet_v2.py:559-565, 602-625 re-associated, DESIGN.md
// section 3: Y = X Wcat^T, 35840 x 12832 x 128 at stage 4) and its like.  1.84 GB of result for 118 GFLOP: these launches are bound
// by their STORES, and on gemm_x3.hip's tile loop (a 256 x 128 tile = 4 chunks of matrix work around a full prologue / epilogue,
// the A tile loaded and split again for each of the 101 column tiles of its row panel) they reach 3.1 TB/s with 1.0 GB fetched
// for 25 MB of operands (profiles/r05_pmc_summary.txt).  Here the ROW PANEL is the unit:
//   * a workgroup (8 waves, two per SIMD) owns 256 rows; wave w holds ITS 32 rows of A -- all K of them, already split into the two
//     scaled fp16 parts of the two-part arithmetic (gemm_x3.hip, NP = 2: same scaling per row, same split, same three partial
//     products in the same order: results are bit-identical to that kernel's) -- in REGISTERS as matrix-instruction fragments
//     (K / 16 k steps x 2 parts x 4 registers), loaded straight from global memory once per panel: no LDS image of A, no
//     conversion work in the loop;
//   * it then sweeps column tiles of 64: the tile of the pre-split weight (two fp16 planes, pdgn_split_f16x2: 64 rows x K x 2
//     parts = 32 KB at K = 128) goes global -> registers -> LDS (the swizzled [row][32 k] image of gemm_x3.hip, double-buffered, one
//     barrier per tile), every wave reads its fragments from there, 3 K / 16 x 2 matrix instructions per wave and tile, and the
//     32 x 64 result leaves through the wave's own 4-KB staging block as whole 128-B lines (gemm_x3.hip's staged stores);
//   * the launch is laid out for the XCDs' L2s: the 8 XCDs form (row groups) x (column slabs) such that a slab of the weight
//     planes stays resident in one XCD's 4-MB L2 (the per-point product: 4 slabs of 1.6 MB), the 32 persistent workgroups of an XCD
//     share its (panels x column tiles) steps in equal contiguous ranges -- a range that crosses into the next panel reloads A.
// Per 64 KB of result a workgroup reads 32 KB of weight planes out of L2 and nothing else; A is read once per column slab.
#include "gemm_shared.h"

typedef _Float16 rp_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rp_f16x2 __attribute__((ext_vector_type(2)));
typedef float rp_f32x16 __attribute__((ext_vector_type(16)));
typedef float rp_f32x2 __attribute__((ext_vector_type(2)));

struct RpArgs {
    long long M;
    int N, lda, ldw, ldc;
    const float *A;
    const unsigned short *Wp;         // two fp16 planes [2][N][ldw], wplane elements apart
    long long wplane;
    float *C;
    const unsigned *max_w;            // row maxima (bit patterns) of W (N: behind the planes); A's are taken in the kernel
    int panels, ctiles;               // row panels of 256, column tiles of 64
    int rgroups, cslabs;              // rgroups * cslabs == 8: XCD x = (row group x / cslabs, column slab x % cslabs)
    int wg_per_xcd;
    float *stat_part;                 // STATS: ceil(M / 256) rows of [3 N] floats, per-column sum (x - pv) | sum (x - pv)^2 | pv of every 256-row panel
                                      // of C (pv: the block's first row), the layout pdgn_bn_stats_from_gemm_partials reads
};

__device__ __forceinline__ unsigned rp_cvt_pk(float a, float b) {
    const rp_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rp_f16x2));
}
__device__ __forceinline__ rp_f32x16 rp_mfma(const u32x4 a, const u32x4 b, const rp_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rp_f16x8, a), __builtin_bit_cast(rp_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned rp_sw(unsigned r) { return (r >> 2) & 3u; }      // gemm_x3.hip x3_sw<32>

template <int K, int AUX, bool PIPE, bool STATS = false>
__global__ __launch_bounds__(512, 1) void gemm_rp_kernel(const RpArgs p) {
    static_assert(K == 32 || K == 64 || K == 128, "reduction lengths of the per-point products");
    constexpr int KS = K / 16, KC = K / 32;                         // k steps of 16; 32-deep chunks of the LDS image
    constexpr int BN = 64, BM = 256;
    constexpr int PLANE = KC * BN * 64;                             // one part of a weight tile: [chunk][64 rows][64 B]
    constexpr int WTILE = 2 * PLANE;                                // h | l
    // LDS: two weight tiles, two tables of the tile's column exponents, eight staging blocks
    // STATS: two tables of the eight waves' partial sums of a tile (64 columns x [sum | squares | pivot]) and their row counts
    constexpr int STL = STATS ? 2 * (8 * 3 * BN + 8) * 4 : 0;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * WTILE + 2 * BN * 4 + 8 * 4096 + STL];
    int *const colexp = reinterpret_cast<int *>(smem + 2 * WTILE);
    float *const stl = reinterpret_cast<float *>(smem + 2 * WTILE + 2 * BN * 4 + 8 * 4096);      // [2][8][3][BN] floats, then [2][8] ints
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *const stg = smem + 2 * WTILE + 2 * BN * 4 + wave * 4096;
    const int li = lane & 31, lg = lane >> 5;

    // this workgroup's share of its XCD's steps
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int rgi = xcd / p.cslabs, csi = xcd - rgi * p.cslabs;
    const int ppg = (p.panels + p.rgroups - 1) / p.rgroups, cps = (p.ctiles + p.cslabs - 1) / p.cslabs;
    const int p0 = rgi * ppg, p1 = min(p.panels, p0 + ppg);
    const int c0 = csi * cps, c1 = min(p.ctiles, c0 + cps);
    if (p0 >= p1 || c0 >= c1) return;
    const int nc = c1 - c0;
    const long long steps = (long long)(p1 - p0) * nc;
    long long t = steps * slot / p.wg_per_xcd;
    const long long t1 = steps * (slot + 1) / p.wg_per_xcd;
    if (t >= t1) return;

    // ---- weight tile: global -> registers -> LDS.  Pieces of 16 B = 8 k of one row of one part; piece id = tid + 512 j:
    // part = id / (BN K / 8), row = (id / (K / 8)) % BN, 16-B column c = id % (K / 8); LDS: chunk c / 4, column (c % 4) ^ sw(row)
    constexpr int PIECES = 2 * BN * (K / 8) / 512;                  // per thread: 4 (K = 128), 2 (64), 1 (32)
    static_assert(PIECES >= 1, "a weight tile is at least one piece per thread");
    u32x4 wraw[PIECES];
    int craw = 0;                                                  // the tile's column exponent this thread fetched (threads 0 .. 63)
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)p.Wp, 0, (int)min((2 * p.wplane) * 2LL, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsMW = __builtin_amdgcn_make_buffer_rsrc((void *)p.max_w, 0, p.N * 4, 0x00020000);
    auto load_w = [&](int ct) {                                    // tile ct of this slab (ct >= nc: nothing)
        const int n0 = (c0 + ct) * BN;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int id = tid + 512 * j;
            const int part = id / (BN * (K / 8)), rem = id - part * (BN * (K / 8));
            const int row = rem / (K / 8), c = rem - row * (K / 8);
            const bool ok = ct < nc && n0 + row < p.N;
            const unsigned off = (unsigned)(((long long)part * p.wplane + (long long)(n0 + row) * p.ldw + 8 * c) * 2);
            wraw[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, ok ? off : NT_OOB, 0, 0));
        }
        if (tid < BN) craw = __builtin_amdgcn_raw_buffer_load_b32(rsMW, (ct < nc && n0 + tid < p.N) ? (unsigned)(n0 + tid) * 4u : NT_OOB, 0, 0);
    };
    auto write_w = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int id = tid + 512 * j;
            const int part = id / (BN * (K / 8)), rem = id - part * (BN * (K / 8));
            const int row = rem / (K / 8), c = rem - row * (K / 8);
            *reinterpret_cast<u32x4 *>(smem + buf * WTILE + part * PLANE + (c >> 2) * (BN * 64) + row * 64 + (((c & 3) ^ rp_sw(row)) * 16)) = wraw[j];
        }
        if (tid < BN) colexp[buf * BN + tid] = 127 - x2_scale_field((unsigned)craw);      // the column's un-scale exponent -e_W
    };

    // ---- A: this wave's 32 rows of the panel as fragments (lane (li, lg) of k step s: row li, k = 16 s + 8 lg .. + 7)
    u32x4 afh[KS], afl[KS];
    int ua_e[4];                                                   // un-scale exponents -e_A of the rows this lane STORES: rows 8 j + (lane >> 3) of the staged read-back
    int cur_mrows = 0;                                             // rows of this wave's block of the current panel inside the matrix
    __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)p.C, 0, 0, 0x00020000);
    auto load_a = [&](int panel) {
        const long long m0 = (long long)panel * BM + 32 * wave;
        const int mrows = (int)max(0LL, min(32LL, p.M - m0));
        cur_mrows = mrows;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(p.A + m0 * p.lda), 0,
                                                                            mrows > 0 ? (int)(((long long)(mrows - 1) * p.lda + K) * 4) : 0, 0x00020000);
        f32x4 raw[2 * KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const unsigned off = li < mrows ? (unsigned)(li * p.lda + 16 * s + 8 * lg) * 4u : NT_OOB;      // (the K tail of the last row must not reach into nothing: whole rows only)
            raw[2 * s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
            raw[2 * s + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off == NT_OOB ? NT_OOB : off + 16u, 0, 0));
        }
        // the row's maximum is taken HERE: the two lanes (li, 0) and (li, 1) hold the whole row between them -- no scan of A in
        // front of the launch, no maxima to hand in (the exact maximum, as a scan finds it: the same scale, the same bits)
        unsigned mb = 0u;
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) mb = max(mb, __float_as_uint(raw[i][r]) & 0x7fffffffu);
        mb = max(mb, (unsigned)__shfl_xor((int)mb, 32));
        const int f = x2_scale_field(mb);
        const float sc = __int_as_float(f << 23);
        // un-scale exponents of the rows this lane STORES (rows 8 j + (lane >> 3) of the staged read-back): from the lanes that hold them
#pragma unroll
        for (int j = 0; j < 4; ++j) ua_e[j] = 127 - __shfl(f, 8 * j + (lane >> 3));
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) {                          // dword e of the fragment: k = 2 e, 2 e + 1 of the lane's eight
                const f32x4 v = raw[2 * s + (e >> 1)];
                const float a2 = v[2 * (e & 1)] * sc, b2 = v[2 * (e & 1) + 1] * sc;
                const unsigned h = rp_cvt_pk(a2, b2);
                const rp_f16x2 hh = __builtin_bit_cast(rp_f16x2, h);
                afh[s][e] = h;
                afl[s][e] = rp_cvt_pk(a2 - (float)hh[0], b2 - (float)hh[1]);
            }
        // the panel's result rows of this wave (rows past M are not written)
        rsC = __builtin_amdgcn_make_buffer_rsrc((void *)(p.C + m0 * p.ldc), 0, mrows > 0 ? (int)(((long long)(mrows - 1) * p.ldc + p.N) * 4) : 0, 0x00020000);
    };

    int panel = p0 + (int)(t / nc), ct = (int)(t - (long long)(panel - p0) * nc);
    load_a(panel);
    load_w(ct);
    write_w(0);
    // (the next tile of the sequence: the same panel's next column tile, or the next panel's first)
    auto next_ct = [&](int c) { return c + 1 < nc ? c + 1 : 0; };
    load_w(t + 1 < t1 ? next_ct(ct) : nc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const unsigned w_rd = (unsigned)(li * 64 + ((lg ^ rp_sw(li)) * 16));       // fragment read: row li of a 32-row block, column 2 (s & 1) + lg of chunk s / 2
    const unsigned stg_wr = (unsigned)(li * 128), stg_sw = (unsigned)(li & 7);
    const unsigned stg_rd = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) * 16));
    // ---- a finished tile's result: out of the scaled domain (exact: 2^-(e_A[row] + e_W[column]) as one ldexp), through the wave's
    // staging block, out as whole 128-B lines.  In four parts -- per 32 x 32 block: accumulators -> staging block; read back row-major
    // (lane l: row 8 j + (l >> 3), columns 4 (l & 7) .. + 3 -- the same four columns for every j: one set of column exponents per
    // block, the row exponents in registers since load_a), un-scale, store -- so that (PIPE) the parts of tile t - 1 sit BETWEEN
    // the matrix instructions of tile t: all eight waves of a workgroup reach the barrier of a step together, and with the result
    // written behind the products every SIMD's matrix pipe stood idle while its two waves stored (measured: 436 us, the products
    // alone 170).
    struct Done {
        rp_f32x16 acc[2];
        i32x4 ec[2];                                               // column exponents of the lane's read-back columns, per block
        int ua[4];
        int n0;
        int mrows;                                                 // rows of this wave's block inside the matrix (0 .. 32)
        int sbuf;                                                  // STATS: which of the two tables takes the tile's partial sums
        __amdgpu_buffer_rsrc_t rsC;
    };
    auto out_part = [&](const Done &d, int part) {
        const int b = part >> 1;
        if ((part & 1) == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<u32x4 *>(stg + stg_wr + (((unsigned)(2 * q + lg) ^ stg_sw) * 16)) =
                    (u32x4){__float_as_uint(d.acc[b][4 * q]), __float_as_uint(d.acc[b][4 * q + 1]), __float_as_uint(d.acc[b][4 * q + 2]), __float_as_uint(d.acc[b][4 * q + 3])};
            return;
        }
        const int nl = d.n0 + 32 * b + 4 * (lane & 7);
        f32x4 val[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(stg + stg_rd + j * 1024);
#pragma unroll
            for (int r = 0; r < 4; ++r) val[j][r] = __builtin_ldexpf(x[r], d.ua[j] + d.ec[b][r]);
            const unsigned off = nl < p.N ? (unsigned)((8 * j + (lane >> 3)) * p.ldc + nl) * 4u : NT_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val[j]), d.rsC, off, 0, AUX);
        }
        if (STATS) {
            // BatchNorm partials of the block's 32 rows, shifted by its first row (gemm_x3.hip's stat_part: sum (x - pv), sum (x - pv)^2,
            // pv; combined in fp64 by cl_finalize_blocks_*).  In the read-back layout a lane holds rows 8 j + (l >> 3) of four columns:
            // four rows in registers, then over the eight lanes of equal l & 7 (xor 8, 16, 32); row 0 sits in lanes 0 .. 7.
            float pv[4], cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] = __shfl(val[0][r], lane & 7);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (8 * j + (lane >> 3) < d.mrows) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dd = val[j][r] - pv[r];
                        cs[r] += dd;
                        cq[r] = __fmaf_rn(dd, dd, cq[r]);
                    }
                }
#pragma unroll
            for (int o = 8; o <= 32; o <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cs[r] += __shfl_xor(cs[r], o);
                    cq[r] += __shfl_xor(cq[r], o);
                }
            // the wave's sums go to the tile's table in LDS: stat_combine joins the eight waves' blocks into ONE partial row per 256-row
            // panel (a partial row per 32 rows made the finalisation of a 358400-row layer walk 11200 rows of [3 N] floats: 69 MB
            // written and read back, two launches of 25-55 us in the step)
            if (lane < 8) {
                float *T = stl + (d.sbuf * 8 + wave) * 3 * BN + 32 * b + 4 * lane;
                *reinterpret_cast<float4 *>(T) = make_float4(cs[0], cs[1], cs[2], cs[3]);
                *reinterpret_cast<float4 *>(T + BN) = make_float4(cq[0], cq[1], cq[2], cq[3]);
                *reinterpret_cast<float4 *>(T + 2 * BN) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                if (b == 0 && lane == 0) reinterpret_cast<int *>(stl + 2 * 8 * 3 * BN)[d.sbuf * 8 + wave] = d.mrows;
            }
        }
    };
    // STATS: the partial row of panel `pan`, columns n0 .. n0 + 63, from table `sb` (complete: a barrier lies between its last write and
    // this call).  Re-based on wave 0's pivot pv0 (the panel's first row): with d = pv_w - pv0,
    //   sum (x - pv0) = S_w + n_w d,   sum (x - pv0)^2 = Q_w + 2 d S_w + n_w d^2      (exact algebra; |d| ~ the column's spread)
    auto stat_combine = [&](int pan, int n0, int sb) {
        if (!STATS || wave != 0 || n0 + lane >= p.N) return;       // (wave 0: BN = 64 columns, one per lane)
        const float *T = stl + sb * 8 * 3 * BN + lane;
        const int *NW = reinterpret_cast<const int *>(stl + 2 * 8 * 3 * BN) + sb * 8;
        const float pv0 = T[2 * BN];
        float S = T[0], Q = T[BN];
#pragma unroll 1                                                   // (one wave's entry at a time: the K = 128 instance has no registers to spare)
        for (int w = 1; w < 8; ++w) {
            const int nw = NW[w];
            if (nw > 0) {
                const float *Tw = T + w * 3 * BN;
                const float sw = Tw[0], qw = Tw[BN], d = Tw[2 * BN] - pv0, fn = (float)nw;
                S += sw + fn * d;
                Q += qw + d * (2.f * sw + fn * d);
            }
        }
        float *P = p.stat_part + (long long)pan * 3 * p.N + n0 + lane;
        P[0] = S;
        P[p.N] = Q;
        P[2 * p.N] = pv0;
    };
    Done prev;
    bool have_prev = false;
    prev.n0 = 0;
    prev.mrows = 0;
    prev.sbuf = 0;
    prev.rsC = rsC;
    int prev_panel = 0;
    // STATS: the tile whose sums the waves wrote during the LAST step (complete since that step's barrier): joined during this one
    bool comb = false;
    int comb_panel = 0, comb_n0 = 0, comb_sb = 0;
    int buf = 0;
    for (; t < t1; ++t) {
        if (STATS && comb) stat_combine(comb_panel, comb_n0, comb_sb);
        // (the tile that leaves during this step -- prev, when there is one -- is joined during the next)
        comb = STATS && have_prev;
        comb_panel = prev_panel;
        comb_n0 = prev.n0;
        comb_sb = prev.sbuf;
        const unsigned char *Wt = smem + buf * WTILE;
        rp_f32x16 acc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        u32x4 wh[2][2], wl[2][2];                                  // [parity of the k step][column block]
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            wh[0][b] = *reinterpret_cast<const u32x4 *>(Wt + w_rd + b * 32 * 64);
            wl[0][b] = *reinterpret_cast<const u32x4 *>(Wt + PLANE + w_rd + b * 32 * 64);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
                const unsigned o = (unsigned)(((s + 1) >> 1) * (BN * 64)) + (w_rd ^ (unsigned)(32 * ((s + 1) & 1)));
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    wh[(s + 1) & 1][b] = *reinterpret_cast<const u32x4 *>(Wt + o + b * 32 * 64);
                    wl[(s + 1) & 1][b] = *reinterpret_cast<const u32x4 *>(Wt + PLANE + o + b * 32 * 64);
                }
            }
            // al wh, ah wl, ah wh -- gemm_x3.hip's order (NP = 2), weight fragment first: lane (li, lg) then holds row li of the
            // block, columns 8 q + 4 lg + (0 .. 3) in registers 4 q .. 4 q + 3
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[b] = rp_mfma(wh[s & 1][b], afl[s], acc[b]);
            // the previous tile's result, a part per k step (KS = 2: two)
            if (PIPE && have_prev) {
#pragma unroll
                for (int part = 0; part < 4; ++part)
                    if ((part * KS) / 4 == s) out_part(prev, part);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[b] = rp_mfma(wl[s & 1][b], afh[s], acc[b]);
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[b] = rp_mfma(wh[s & 1][b], afh[s], acc[b]);
        }
        // this tile leaves during the next step's products (PIPE), or now
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            prev.acc[b] = acc[b];
            prev.ec[b] = *reinterpret_cast<const i32x4 *>(colexp + buf * BN + 32 * b + 4 * (lane & 7));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) prev.ua[j] = ua_e[j];
        prev.n0 = (c0 + ct) * BN;
        prev.mrows = cur_mrows;
        prev.sbuf = (int)(t & 1);
        prev_panel = panel;
        prev.rsC = rsC;
        have_prev = true;
        if (!PIPE) {
#pragma unroll
            for (int part = 0; part < 4; ++part) out_part(prev, part);
        }
        // the next tile's planes (in registers since the last step) into the other buffer: every wave left that buffer at the
        // barrier that ended the previous step; then the loads of the tile after next
        const bool more = t + 1 < t1;
        const int nct = next_ct(ct);
        if (more) write_w(buf ^ 1);
        const bool more2 = t + 2 < t1;
        if (more2) load_w(next_ct(nct));
        // next step: the same panel's next tile, or the next panel
        if (more && nct == 0) {
            ++panel;
            load_a(panel);
        }
        ct = nct;
        buf ^= 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (STATS && comb) stat_combine(comb_panel, comb_n0, comb_sb);
    if (PIPE && have_prev) {
#pragma unroll
        for (int part = 0; part < 4; ++part) out_part(prev, part);
    }
    if (STATS && have_prev) {                                       // the last tile's sums: written just now
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stat_combine(prev_panel, prev.n0, prev.sbuf);
    }
}

// ------------------------------------------------------------------ host side
template <int K>
static void rp_go(int grid, hipStream_t s, const RpArgs &a, int aux, bool pipe) {
    if (a.stat_part) {
        if (aux == 2) hipLaunchKernelGGL((gemm_rp_kernel<K, 2, true, true>), dim3(grid), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((gemm_rp_kernel<K, 0, true, true>), dim3(grid), dim3(512), 0, s, a);
    } else if (!pipe) hipLaunchKernelGGL((gemm_rp_kernel<K, 2, false>), dim3(grid), dim3(512), 0, s, a);
    else if (aux == 16) hipLaunchKernelGGL((gemm_rp_kernel<K, 16, true>), dim3(grid), dim3(512), 0, s, a);
    else if (aux == 2) hipLaunchKernelGGL((gemm_rp_kernel<K, 2, true>), dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm_rp_kernel<K, 0, true>), dim3(grid), dim3(512), 0, s, a);
}

// Whether pdgn_gemm_nt_ps(m, n, k) with two-part planes and a plain epilogue takes this kernel: a short reduction, a result at
// least ~64 MB wide enough that the stores are what the launch costs.  PDGN_RP=0: never (A/B switch).
bool rp_takes(long long m, int n, int k) {
    static const bool off = [] { const char *e = getenv("PDGN_RP"); return e && e[0] == '0'; }();
    return !off && (k == 32 || k == 64 || k == 128) && n >= 128 && m >= 4096 && (double)m * n >= 1.6e7 && m < (1LL << 28);
}

// C (m x n, pitch ldc) = A (m x k, pitch lda) W^T for two-part planes Wp [2][n][ldw] (pdgn_split_f16x2: the rows' maxima behind
// them); A's row maxima are taken in the kernel (a wave holds whole rows).  Returns 0, or a launch error.
int rp_launch(long long m, int n, int k, const float *A, int lda, const unsigned short *Wp, int ldw, long long wplane, float *C, int ldc,
              float *stat_part, hipStream_t s) {
    RpArgs a;
    a.stat_part = stat_part;
    a.M = m; a.N = n; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.A = A; a.Wp = Wp; a.wplane = wplane; a.C = C;
    a.max_w = reinterpret_cast<const unsigned *>(Wp + 2 * wplane);
    a.panels = cdiv(m, 256);
    a.ctiles = cdiv(n, 64);
    // column slabs: the planes of a slab (tiles x 64 rows x k x 2 parts x 2 B) should stay in an XCD's 4-MB L2 beside the streams
    static const int force_cs = [] { const char *e = getenv("PDGN_RP_CSLABS"); return e ? atoi(e) : 0; }();
    int cs = 1;
    while (cs < 8 && (double)cdiv(a.ctiles, cs) * 64 * k * 4 > 2.0e6) cs *= 2;
    if (force_cs == 1 || force_cs == 2 || force_cs == 4 || force_cs == 8) cs = force_cs;
    while (cs > 1 && (a.ctiles < cs || a.panels < 8 / cs)) cs /= 2;                    // (every XCD gets rows and columns)
    a.cslabs = cs;
    a.rgroups = 8 / cs;
    const int cus = nt_cus();
    a.wg_per_xcd = cus >= 8 ? cus / 8 : 1;
    // store policy: a result beyond what the 256-MB Infinity Cache keeps is streamed (nt: 447 -> 422 us at 1.84 GB); a smaller one is
    // left where its consumer -- the gather kernel that follows -- finds it (plain: 49.9 -> 40.6 us at 232 MB).  PDGN_RP_STORE = 0 | 2 | 16
    // forces plain / nt / sc1 (measurement)
    static const int aux_env = [] { const char *e = getenv("PDGN_RP_STORE"); return e ? atoi(e) : -1; }();
    const int aux = aux_env >= 0 ? aux_env : ((double)m * n * 4 > 2.5e8 ? 2 : 0);
    // (unstaged stores -- 32 B of each of 32 rows per instruction -- measured 510 us against 436 staged, and 2207 with nt: partial lines
    // written around the L2 are read-modify-writes at the memory; not built)
    static const bool pipe = [] { const char *e = getenv("PDGN_RP_PIPE"); return !(e && e[0] == '0'); }();   // 0: a tile's result behind its products (A/B)
    const int grid = 8 * a.wg_per_xcd;
    if (k == 128) rp_go<128>(grid, s, a, aux, pipe);
    else if (k == 64) rp_go<64>(grid, s, a, aux, pipe);
    else rp_go<32>(grid, s, a, aux, pipe);
    return pdgn_launch_status();
}
