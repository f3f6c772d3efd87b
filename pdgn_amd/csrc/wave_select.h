// wave_select.h -- wave64 top-K selection primitives shared by the 3-D kNN (knn3.hip) and the
// feature-space kNN (feat_knn.hip): bitonic sorts over ds_bpermute and the survivor-queue merge.
#pragma once
#include "common.h"
#ifndef WSEL_T
#define WSEL_T(n) do { } while (0)        // timing hook of experiment builds
#endif

struct DI {
    float d;
    int i;
};

__device__ __forceinline__ bool di_less(float d0, int i0, float d1, int i1) {
    return d0 < d1 || (d0 == d1 && i0 < i1);
}

// Wave-wide bitonic sort (ascending) of one (d, i) pair per lane.
__device__ __forceinline__ void wave_sort_di(float &d, int &i, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            float pd = __shfl_xor(d, j, 64);
            int pi = __shfl_xor(i, j, 64);
            bool keep_min = ((lane & j) == 0) == ((lane & k2) == 0);
            bool partner_less = di_less(pd, pi, d, i);
            bool take = keep_min ? partner_less : !partner_less;
            d = take ? pd : d;
            i = take ? pi : i;
        }
    }
}

// Wave-wide bitonic sort (ascending) of one float per lane.
__device__ __forceinline__ float wave_sort_f(float d, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            float pd = __shfl_xor(d, j, 64);
            bool keep_min = ((lane & j) == 0) == ((lane & k2) == 0);
            d = keep_min ? fminf(d, pd) : fmaxf(d, pd);
        }
    }
    return d;
}

// Merge the queued survivors q[0..cnt) into the running best-K (lanes 0..K-1 hold it, sorted).
__device__ __forceinline__ void knn_flush(const DI *q, int cnt, int K, float &rd, int &ri, int lane) {
    const int take = 64 - K;
    __builtin_amdgcn_wave_barrier();            // queue writes of other lanes precede these reads
    for (int base = 0; base < cnt; base += take) {
        int src = base + lane - K;
        bool fresh = lane >= K;
        float d = rd;
        int i = ri;
        if (fresh) {
            bool ok = src < cnt;
            d = ok ? q[ok ? src : 0].d : INFINITY;
            i = ok ? q[ok ? src : 0].i : 0x7fffffff;
        }
        wave_sort_di(d, i, lane);
        rd = d;
        ri = i;
    }
    __builtin_amdgcn_wave_barrier();
}


#define WSEL_QCAP 256   // survivor queue entries per wave

__device__ __forceinline__ float wave_kth_smallest(float v, int K);

// One wave scans `tn` candidates (candidate c has distance dist(c), c in [0, tn), global index
// t0 + c) and merges the K best into the running list (rd, ri) held by lanes 0..K-1.
//   pass A: per-lane minimum; tau = K-th smallest lane minimum (>= the K-th nearest distance);
//   pass B: survivors d <= tau are compacted into the wave's LDS queue `q` and merged.
// +inf distances never enter (the reference's `d2 < best` against 1e40 never accepts them).
template <class DistFn>
__device__ __forceinline__ void wave_topk_scan(DistFn dist, int tn, int t0, DI *q, int K, float &rd,
                                               int &ri, int lane) {
    float lmin = INFINITY;
    for (int c = lane; c < tn; c += 64) lmin = fminf(lmin, dist(c));
    float tau = fminf(wave_kth_smallest(lmin, K), __shfl(rd, K - 1, 64));   // ballot bisection, no LDS
    int cnt = 0;                                    // wave-uniform
    for (int c0 = 0; c0 < tn; c0 += 64) {
        int c = c0 + lane;
        float d = c < tn ? dist(c) : INFINITY;
        bool keep = d <= tau && d < INFINITY;
        unsigned long long mask = __ballot(keep);
        if (mask) {
            int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (keep) { q[pos].d = d; q[pos].i = t0 + c; }
            cnt += __popcll(mask);
            if (cnt > WSEL_QCAP - 64) {
                knn_flush(q, cnt, K, rd, ri, lane);
                cnt = 0;
                tau = fminf(tau, __shfl(rd, K - 1, 64));
            }
        }
    }
    knn_flush(q, cnt, K, rd, ri, lane);
}

// K-th smallest (1-based) of the 64 lane values, exact, by bisection on the order-preserving
// integer image of the floats: 32 ballot/popcount steps on scalar registers, no LDS traffic
// (the bitonic wave_sort_f costs 21 dependent ds_bpermute round trips).
__device__ __forceinline__ float wave_kth_smallest(float v, int K) {
    const unsigned bits = __float_as_uint(v);
    const unsigned key = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
    unsigned prefix = 0;
#pragma unroll
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = prefix | (1u << bit);
        const int below = __popcll(__ballot(key < cand));
        prefix = below < K ? cand : prefix;
    }
    const unsigned b = (prefix & 0x80000000u) ? (prefix ^ 0x80000000u) : ~prefix;
    return __uint_as_float(b);
}

// Upper bound of the K-th smallest lane value: the bisection stops after the top 16 bits of the order-preserving
// image and rounds the rest up (within 0.8 % of the exact value for the distances selected here -- as good a filter
// threshold as the exact one at half the dependent steps).  Any sign; never past +inf.
__device__ __forceinline__ float wave_kth_smallest_ub16_any(float v, int K) {
    const unsigned bits = __float_as_uint(v);
    const unsigned key = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
    unsigned prefix = 0;
#pragma unroll
    for (int bit = 31; bit >= 16; --bit) {
        const unsigned cand = prefix | (1u << bit);
        const int below = __popcll(__ballot(key < cand));
        prefix = below < K ? cand : prefix;
    }
    const unsigned ub = min(prefix | 0xffffu, 0xff800000u);
    return __uint_as_float((ub & 0x80000000u) ? (ub ^ 0x80000000u) : ~ub);
}

// Variant of wave_topk_scan with a per-query survivor queue that persists across candidate
// chunks: survivors are only appended here (threshold from this chunk's lane minima and the
// running list); the (distance, index) sort runs when the queue is half full or at the end.
#define WSEL_PQCAP 128
template <class DistFn>
__device__ __forceinline__ void wave_topk_append(DistFn dist, int tn, int t0, DI *q, int &cnt, int K,
                                                 float &rd, int &ri, int lane) {
    float lmin = INFINITY;
    for (int c = lane; c < tn; c += 64) lmin = fminf(lmin, dist(c));
    float tau = fminf(wave_kth_smallest(lmin, K), __shfl(rd, K - 1, 64));
    for (int c0 = 0; c0 < tn; c0 += 64) {
        if (cnt > WSEL_PQCAP - 64) {
            knn_flush(q, cnt, K, rd, ri, lane);
            cnt = 0;
            tau = fminf(tau, __shfl(rd, K - 1, 64));
        }
        int c = c0 + lane;
        float d = c < tn ? dist(c) : INFINITY;
        bool keep = d <= tau && d < INFINITY;
        unsigned long long mask = __ballot(keep);
        if (mask) {
            int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (keep) { q[pos].d = d; q[pos].i = t0 + c; }
            cnt += __popcll(mask);
        }
    }
}

// As wave_topk_append, for a queue of `cap` >= 64 entries: the queue is merged into the running list
// only when the next 64-candidate step would not fit (wave-uniform test on the exact count).
template <class DistFn>
__device__ __forceinline__ void wave_topk_append_cap(DistFn dist, int tn, int t0, DI *q, int &cnt, int cap, int K,
                                                     float &rd, int &ri, int lane) {
    float lmin = INFINITY;
    for (int c = lane; c < tn; c += 64) lmin = fminf(lmin, dist(c));
    float tau = fminf(wave_kth_smallest_ub16_any(lmin, K), __shfl(rd, K - 1, 64));
    for (int c0 = 0; c0 < tn; c0 += 64) {
        int c = c0 + lane;
        float d = c < tn ? dist(c) : INFINITY;
        bool keep = d <= tau && d < INFINITY;
        unsigned long long mask = __ballot(keep);
        if (mask) {
            const int add = __popcll(mask);
            if (cnt + add > cap) {
                knn_flush(q, cnt, K, rd, ri, lane);
                cnt = 0;
                tau = fminf(tau, __shfl(rd, K - 1, 64));
                keep = keep && d <= tau;                      // tightened threshold: re-filter this step
                mask = __ballot(keep);
            }
            int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (keep) { q[pos].d = d; q[pos].i = t0 + c; }
            cnt += __popcll(mask);
        }
    }
}

// wave_topk_append_cap for Q queries at once (rows[t] = LDS row of query t, tn a multiple of 64): the LDS reads, the
// threshold bisections and the ballots of the Q queries are independent chains the hardware overlaps -- one query at a
// time the pass is a sequence of exposed LDS / VALU->SALU latencies.  `flush(t)` merges query t's full queue into its
// running list (rd[t], ri[t]) and empties it.  Bit t of `active` clear: slot t has no query (its row pointer must still
// be readable); nothing is queued for it.
template <int Q, class FlushFn>
__device__ __forceinline__ void wave_topk_append_multi(const float *const (&rows)[Q], int tn, int t0, DI *const (&q)[Q],
                                                       int (&cnt)[Q], int cap, int K, float (&rd)[Q], int (&ri)[Q], int lane,
                                                       FlushFn flush, unsigned active = ~0u) {
    WSEL_T(6);
    // the counters are wave-uniform: say so, and the compiler keeps them (and the overflow test) on the scalar unit
    int sc[Q];
    const float *lr[Q];                                     // this lane's column of each row: the steps are immediates
#pragma unroll
    for (int t = 0; t < Q; ++t) { sc[t] = __builtin_amdgcn_readfirstlane(cnt[t]); lr[t] = rows[t] + lane; }
    float lmin[Q];
#pragma unroll
    for (int t = 0; t < Q; ++t) lmin[t] = INFINITY;
#pragma unroll
    for (int c0 = 0; c0 < 512; c0 += 64) {
        if (c0 >= tn) break;                                // wave-uniform (tn <= 512, a multiple of 64)
#pragma unroll
        for (int t = 0; t < Q; ++t) lmin[t] = fminf(lmin[t], lr[t][c0]);
    }
    WSEL_T(7);
    unsigned key[Q], prefix[Q];
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const unsigned bits = __float_as_uint(lmin[t]);
        key[t] = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
        prefix[t] = 0;
    }
#pragma unroll
    for (int bit = 31; bit >= 16; --bit) {                // the Q chains stage by stage: compares, counts, selects
        unsigned long long lo[Q];
#pragma unroll
        for (int t = 0; t < Q; ++t) lo[t] = __ballot(key[t] < (prefix[t] | (1u << bit)));
#pragma unroll
        for (int t = 0; t < Q; ++t)
            prefix[t] = __builtin_amdgcn_readfirstlane(__popcll(lo[t]) < K ? (prefix[t] | (1u << bit)) : prefix[t]);
    }
    WSEL_T(8);
    float tau[Q];
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const unsigned ub = min(prefix[t] | 0xffffu, 0xff800000u);
        tau[t] = fminf(__uint_as_float((ub & 0x80000000u) ? (ub ^ 0x80000000u) : ~ub), __shfl(rd[t], K - 1, 64));
        tau[t] = fminf(tau[t], 3.402823466e38f);            // finite: "d <= tau" alone keeps +inf out of the queue
        if (!((active >> t) & 1u)) tau[t] = -3.402823466e38f;   // slot without a query (wave-uniform): nothing passes
    }
    WSEL_T(9);
    // Pass B, written for few VALU -> SALU -> VALU round trips (each costs tens of cycles, more next to a matrix
    // stream that owns the vector issue slots): the Q compares, then the Q slot computations (mbcnt on the masks), ONE
    // scalar overflow test for the step, then the predicated writes.  No per-query "any survivor?" branch.
#pragma unroll
    for (int c0 = 0; c0 < 512; c0 += 64) {
        if (c0 >= tn) break;
        float d[Q];
        unsigned long long mask[Q];
        int add[Q];
#pragma unroll
        for (int t = 0; t < Q; ++t) d[t] = lr[t][c0];
#pragma unroll
        for (int t = 0; t < Q; ++t) mask[t] = __ballot(d[t] <= tau[t]);
        bool over = false;
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            add[t] = __popcll(mask[t]);
            over = over || sc[t] + add[t] > cap;
        }
        const int ci = t0 + c0 + lane;
        if (!over) {
#pragma unroll
            for (int t = 0; t < Q; ++t) {
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask[t] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask[t], sc[t]));
                if (d[t] <= tau[t]) { q[t][pos].d = d[t]; q[t][pos].i = ci; }
                sc[t] += add[t];
            }
        } else {                                             // a queue would overflow: that query merges first (rare)
#pragma unroll
            for (int t = 0; t < Q; ++t) {
                bool keep = d[t] <= tau[t];
                unsigned long long mk = mask[t];
                if (sc[t] + add[t] > cap) {
                    cnt[t] = sc[t];
                    flush(t);
                    sc[t] = __builtin_amdgcn_readfirstlane(cnt[t]);
                    tau[t] = fminf(tau[t], __shfl(rd[t], K - 1, 64));
                    keep = keep && d[t] <= tau[t];            // tightened threshold: re-filter this step
                    mk = __ballot(keep);
                }
                const int p = sc[t] + __popcll(mk & ((1ull << lane) - 1ull));
                if (keep) { q[t][p].d = d[t]; q[t][p].i = ci; }
                sc[t] += __popcll(mk);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < Q; ++t) cnt[t] = sc[t];
    WSEL_T(10);
}

// Rank-based merge (no sorting network): the running list's valid entries are appended to the queue,
// every entry computes its rank = #entries smaller in the (distance, index) total order by streaming
// the queue through broadcast LDS reads (independent loads, no dependent cross-lane chain), and the
// entries of rank < K drop into win[rank].  Replaces the 21-stage bitonic sort of knn_flush (each stage
// two dependent ds_bpermute round trips) when a K-entry LDS scratch `win` is available.
__device__ __forceinline__ void knn_flush_ranked(DI *q, int cnt, int K, float &rd, int &ri, int lane, DI *win) {
    __builtin_amdgcn_wave_barrier();
    const bool have = lane < K && rd < INFINITY;
    const unsigned long long hm = __ballot(have);
    if (have) {
        const int pos = cnt + __popcll(hm & ((1ull << lane) - 1ull));
        q[pos].d = rd; q[pos].i = ri;
    }
    cnt += __popcll(hm);
    if (lane < K) { win[lane].d = INFINITY; win[lane].i = 0x7fffffff; }
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < cnt; base += 64) {
        const int me = base + lane;
        const bool ok = me < cnt;
        const float d = ok ? q[me].d : INFINITY;
        const int i = ok ? q[me].i : 0x7fffffff;
        int rank = 0;
        int j = 0;
        for (; j + 8 <= cnt; j += 8) {               // eight broadcast reads in flight: the loop is LDS latency otherwise
            float dj[8];
            int ij[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { dj[u] = q[j + u].d; ij[u] = q[j + u].i; }
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (dj[u] < d || (dj[u] == d && ij[u] < i)) ? 1 : 0;
        }
        for (; j < cnt; ++j) {
            const float dj = q[j].d;
            const int ij = q[j].i;
            rank += (dj < d || (dj == d && ij < i)) ? 1 : 0;
        }
        if (ok && rank < K) { win[rank].d = d; win[rank].i = i; }
    }
    __builtin_amdgcn_wave_barrier();
    rd = lane < K ? win[lane].d : INFINITY;
    ri = lane < K ? win[lane].i : 0x7fffffff;
    __builtin_amdgcn_wave_barrier();
}

// Final merge by selection instead of ranking everything: fp32 MFMA and the vector ALU share a SIMD's issue slots on
// gfx950 (the matrix path IS the vector rate), so next to a Gram producer every VALU instruction of the selection waves is
// time taken from the matrix pipe -- knn_flush_ranked spends ~4 VALU instructions per (entry, entry) pair of the queue.
// Here the exact K-th smallest distance is bisected over the (<= 128) queued entries by ballots (2 compares per bit, the
// counting is scalar), the entries up to it -- K of them unless distances tie at the threshold -- are compacted into
// `win`, and only those are ranked in the (distance, index) order.  More than 32 entries at or below the K-th distance
// (mass ties) or more than 128 queued entries: the ranked merge does it.
__device__ __forceinline__ void knn_flush_select(DI *q, int cnt, int K, float &rd, int &ri, int lane, DI *win) {
    __builtin_amdgcn_wave_barrier();
    const bool have = lane < K && rd < INFINITY;
    const unsigned long long hm = __ballot(have);
    const int total = cnt + __popcll(hm);
    if (total > 128) {
        knn_flush_ranked(q, cnt, K, rd, ri, lane, win);
        return;
    }
    if (have) {
        const int pos = cnt + __popcll(hm & ((1ull << lane) - 1ull));
        q[pos].d = rd; q[pos].i = ri;
    }
    __builtin_amdgcn_wave_barrier();
    const bool ok0 = lane < total, ok1 = lane + 64 < total;
    const float d0 = ok0 ? q[lane].d : INFINITY, d1 = ok1 ? q[lane + 64].d : INFINITY;
    const int i0 = ok0 ? q[lane].i : 0x7fffffff, i1 = ok1 ? q[lane + 64].i : 0x7fffffff;
    const unsigned b0 = __float_as_uint(d0), b1 = __float_as_uint(d1);
    const unsigned k0 = b0 ^ ((b0 >> 31) ? 0xffffffffu : 0x80000000u), k1 = b1 ^ ((b1 >> 31) ? 0xffffffffu : 0x80000000u);
    unsigned prefix = 0;
#pragma unroll
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = prefix | (1u << bit);
        const int below = __popcll(__ballot(k0 < cand)) + __popcll(__ballot(k1 < cand));
        prefix = below < K ? cand : prefix;
    }
    // prefix = key of the K-th smallest distance (of +inf when fewer than K entries are finite)
    const bool in0 = k0 <= prefix && d0 < INFINITY, in1 = k1 <= prefix && d1 < INFINITY;
    const unsigned long long m0 = __ballot(in0), m1 = __ballot(in1);
    const int n0 = __popcll(m0), m = n0 + __popcll(m1);
    if (m > 32) {                                   // wave-uniform: mass ties at the threshold
        knn_flush_ranked(q, total, K, rd = INFINITY, ri = 0x7fffffff, lane, win);
        return;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (in0) { const int p = __popcll(m0 & lt); win[p].d = d0; win[p].i = i0; }
    if (in1) { const int p = n0 + __popcll(m1 & lt); win[p].d = d1; win[p].i = i1; }
    __builtin_amdgcn_wave_barrier();
    int r0 = 0, r1 = 0;
    for (int j = 0; j < m; ++j) {
        const float dj = win[j].d;
        const int ij = win[j].i;
        r0 += (dj < d0 || (dj == d0 && ij < i0)) ? 1 : 0;
        r1 += (dj < d1 || (dj == d1 && ij < i1)) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();                // every lane has read win before q (its alias in some callers) is rewritten
    if (in0 && r0 < K) { q[r0].d = d0; q[r0].i = i0; }
    if (in1 && r1 < K) { q[r1].d = d1; q[r1].i = i1; }
    __builtin_amdgcn_wave_barrier();
    const bool out = lane < K && lane < m;
    rd = out ? q[lane].d : INFINITY;
    ri = out ? q[lane].i : 0x7fffffff;
    __builtin_amdgcn_wave_barrier();
}

// knn_flush_select for Q queries at once: the same steps with the Q dependency chains (LDS round trips, VALU -> SALU ->
// VALU bisection steps) interleaved stage by stage; queries that need the general path (see above) are finished one by
// one afterwards.  Differences to the single-query form, all for fewer dependent steps: the threshold is the 16-bit upper
// bound of the K-th smallest distance (a few entries more than K pass it; the ranking sorts that out exactly), and the
// compacted survivors are read back one per lane so that the ranking loop broadcasts them with v_readlane instead of LDS
// reads.  The entries are in registers before anything is overwritten, so the compaction reuses the head of each queue;
// `scratch` (32 entries) only serves the general path.
template <int Q>
__device__ __forceinline__ void knn_flush_select_multi(DI *const (&q)[Q], int (&cnt)[Q], int K, float (&rd)[Q], int (&ri)[Q],
                                                       int lane, DI *scratch) {
    WSEL_T(0);
    __builtin_amdgcn_wave_barrier();
    int total[Q];
    bool slow[Q];
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const bool have = lane < K && rd[t] < INFINITY;
        const unsigned long long hm = __ballot(have);
        total[t] = __builtin_amdgcn_readfirstlane(cnt[t] + __popcll(hm));       // wave-uniform: keep the chain scalar
        slow[t] = total[t] > 128;
        if (have && !slow[t]) {
            const int pos = cnt[t] + __popcll(hm & ((1ull << lane) - 1ull));
            q[t][pos].d = rd[t]; q[t][pos].i = ri[t];
        }
    }
    __builtin_amdgcn_wave_barrier();
    float d0[Q], d1[Q];
    int i0[Q], i1[Q];
    unsigned k0[Q], k1[Q], prefix[Q];
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const bool ok0 = lane < total[t], ok1 = lane + 64 < total[t];
        d0[t] = ok0 ? q[t][lane].d : INFINITY;
        d1[t] = ok1 ? q[t][lane + 64].d : INFINITY;
        i0[t] = ok0 ? q[t][lane].i : 0x7fffffff;
        i1[t] = ok1 ? q[t][lane + 64].i : 0x7fffffff;
    }
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const unsigned b0 = __float_as_uint(d0[t]), b1 = __float_as_uint(d1[t]);
        k0[t] = b0 ^ ((b0 >> 31) ? 0xffffffffu : 0x80000000u);
        k1[t] = b1 ^ ((b1 >> 31) ? 0xffffffffu : 0x80000000u);
        prefix[t] = 0;
    }
    WSEL_T(1);
#pragma unroll
    for (int bit = 31; bit >= 16; --bit) {
        unsigned long long lo[Q], hi[Q];
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            const unsigned cand = prefix[t] | (1u << bit);
            lo[t] = __ballot(k0[t] < cand);
            hi[t] = __ballot(k1[t] < cand);
        }
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            const int below = __popcll(lo[t]) + __popcll(hi[t]);
            prefix[t] = __builtin_amdgcn_readfirstlane(below < K ? (prefix[t] | (1u << bit)) : prefix[t]);
        }
    }
    WSEL_T(2);
    bool in0[Q], in1[Q];
    int m[Q];
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const unsigned ub = prefix[t] | 0xffffu;            // >= the K-th smallest key
        in0[t] = k0[t] <= ub && d0[t] < INFINITY;
        in1[t] = k1[t] <= ub && d1[t] < INFINITY;
        const unsigned long long m0 = __ballot(in0[t]), m1 = __ballot(in1[t]);
        const int n0 = __popcll(m0);
        m[t] = __builtin_amdgcn_readfirstlane(n0 + __popcll(m1));
        slow[t] = slow[t] || m[t] > 32;
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (!slow[t]) {
            if (in0[t]) { const int p = __popcll(m0 & lt); q[t][p].d = d0[t]; q[t][p].i = i0[t]; }
            if (in1[t]) { const int p = n0 + __popcll(m1 & lt); q[t][p].d = d1[t]; q[t][p].i = i1[t]; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // lane p < m holds survivor p; its rank among the survivors is its place in the output.  (distance, index) as ONE
    // 64-bit key: a v_cmp_lt_u64 and an add per pair, no scalar logic in the loop.
    float sd[Q];
    int si[Q], rank[Q], mmax = 0;
    unsigned long long sk[Q];
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const bool ok = !slow[t] && lane < m[t];
        sd[t] = ok ? q[t][lane & 31].d : INFINITY;
        si[t] = ok ? q[t][lane & 31].i : 0x7fffffff;
        const unsigned b = __float_as_uint(sd[t]);
        sk[t] = ((unsigned long long)(b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u)) << 32) | (unsigned)si[t];
        rank[t] = 0;
        mmax = max(mmax, slow[t] ? 0 : m[t]);
    }
    WSEL_T(3);
#pragma unroll
    for (int j = 0; j < 32; ++j) {                         // constant lane selects; lanes >= m hold the largest key: they rank nothing
        if (j >= mmax) break;                              // wave-uniform
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            const unsigned hi = __builtin_amdgcn_readlane((unsigned)(sk[t] >> 32), j);
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)sk[t], j);
            rank[t] += ((((unsigned long long)hi << 32) | lo) < sk[t]) ? 1 : 0;
        }
    }
    WSEL_T(4);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        if (!slow[t] && lane < m[t] && rank[t] < K) { q[t][rank[t]].d = sd[t]; q[t][rank[t]].i = si[t]; }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        if (!slow[t]) {
            const bool out = lane < K && lane < m[t];
            rd[t] = out ? q[t][lane].d : INFINITY;
            ri[t] = out ? q[t][lane].i : 0x7fffffff;
        }
    }
    __builtin_amdgcn_wave_barrier();
    WSEL_T(5);
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        if (slow[t]) {                                   // wave-uniform, rare
            if (total[t] > 128) {
                knn_flush_ranked(q[t], cnt[t], K, rd[t], ri[t], lane, scratch);
            } else {
                float none_d = INFINITY;
                int none_i = 0x7fffffff;
                knn_flush_ranked(q[t], total[t], K, none_d, none_i, lane, scratch);
                rd[t] = none_d; ri[t] = none_i;
            }
        }
    }
}
