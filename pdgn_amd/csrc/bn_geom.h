// bn_geom.h -- launch geometry shared by the channels-last BatchNorm reductions (bnact.hip) and by the producers
// that emit the same per-row-block partial sums in their epilogue (wgs.hip).
#pragma once
#define BN_THREADS 256

// CG = C/4 float4 column groups; a block owns `cgb` of them and rows_per_block rows; thread t owns column group
// t % cgb and row lane t / cgb.  part[row_block][c] / [C + c] are the fp32 partial sum / sum of squares.
static inline void cl_geometry(long long R, int C, int *cgb, int *gx, int *gy, int *rpb) {
    const int cg = C / 4;
    int p = 1;                                             // column groups per block: power of two <= 256
    while (p < cg && p < BN_THREADS) p <<= 1;
    *cgb = p;
    *gx = (cg + *cgb - 1) / *cgb;
    const int rl = BN_THREADS / *cgb;
    long long want = 1024 / *gx;                           // ~1024 workgroups in flight
    want = want < 1 ? 1 : want;
    long long rows = (R + want - 1) / want;
    const long long min_rows = (long long)rl * 16;
    rows = rows < min_rows ? min_rows : rows;
    rows = rows > 65536 ? 65536 : rows;                    // bound the fp32 partial sums
    rows = (rows + rl - 1) / rl * rl;
    *rpb = (int)rows;
    *gy = (int)((R + rows - 1) / rows);
}

