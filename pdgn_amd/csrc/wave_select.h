// wave_select.h -- wave64 top-K selection primitives shared by the 3-D kNN (knn3.hip) and the
// feature-space kNN (feat_knn.hip): bitonic sorts over ds_bpermute and the survivor-queue merge.
#pragma once
#include "common.h"

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
        for (int j = 0; j < cnt; ++j) {
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
