// knn3.hip -- brute-force 3-D kNN (pdgn_knnquery) and 3-NN (pdgn_nearestneighbor) for gfx950.
//
// Semantics: lib/pointops/src/knnquery/knnquery_cuda_kernel.cu:6-50 and
// interpolation/interpolation_cuda_kernel.cu:134-176 of the reference -- ascending
// (squared distance, index), strict '<', tail idx 0 / +inf when the set is short.
//
// Design (not the reference's one-thread-per-query insertion sort): a wave serves a GROUP of queries (knn3_wave4_kernel).
//   * the candidate set of a batch is staged once per workgroup into LDS as float4, padded with points at infinity
//     (coalesced global reads, guard-free scans, one ds_read_b128 of a candidate feeds every query of the group);
//   * pass A: lane l scans candidates l, l+64, ... and keeps only its minimum per query; a 16-bit ballot bisection over
//     the 64 lane minima gives an upper bound tau of the K-th nearest distance;
//   * pass B: lanes recompute their distances and compact the survivors (d <= tau) into per-query LDS queues with
//     ballot / popcount prefix sums (typically 25-40 survivors);
//   * the survivors are ranked by (distance, index) (broadcast LDS reads, no cross-lane chain) and written in order.
// (distance, index) is a strict total order, so the result is bit-identical to the reference's stable insertion
// regardless of scheduling.  k > 32 (the reference allows 200) runs on knn3_generic_kernel, one thread per query.
#include "common.h"
#include "wave_select.h"

#define KNN_THREADS 256
#define KNN_WAVES (KNN_THREADS / PDGN_WAVE)
#define KNN_TILE 4096        // candidates staged per pass: 64 KiB of float4
#define KNN_FAST_MAX_K 32


// ---------------------------------------------------------------------------------------------------------------
// Several queries per wave (round 2; KNN4_Q = 2 measured fastest: 4 halves the occupancy, 1 the reuse).  The one-query kernel above spends most of a query's ~5000 cycles in dependent chains
// (32 ballot-bisection steps for the threshold, a 21-stage bitonic sort over ds_bpermute for the merge) at two waves
// per SIMD.  Here a wave owns KNN4_Q consecutive queries: one ds_read_b128 of a candidate feeds their distance chains, the
// threshold searches / survivor queues / merges are independent instruction streams the scheduler interleaves, the
// threshold search stops after 16 bits (an upper bound within 0.8 % of the exact K-th lane minimum is as good a filter),
// the merge is the rank-based one of wave_select.h (broadcast LDS reads, no cross-lane chain), the candidate tile is
// staged 4 points = three float4 per thread, and the workgroup is sized for six waves per SIMD (48 KB of LDS).
// Results are bit-identical: same sqdist3 chain, same strict (distance, index) order.
#define KNN4_THREADS 512
#define KNN4_WAVES (KNN4_THREADS / PDGN_WAVE)
#define KNN4_Q 2
#define KNN4_TILE 2048       // candidates staged per pass: 32 KiB of float4
#define KNN4_QCAP 64         // survivor queue entries per query (+ room for the running list in a merge)

// Upper bound of the K-th smallest (1-based) of the 64 lane values (all >= 0 or +inf): bisection over the top 16 bits of
// the order-preserving integer image, the rest rounded up.
__device__ __forceinline__ float wave_kth_smallest_ub16(float v, int K) {
    const unsigned key = __float_as_uint(v) | 0x80000000u;       // v >= +0: image = bits ^ 0x80000000
    unsigned prefix = 0x80000000u;
#pragma unroll
    for (int bit = 30; bit >= 15; --bit) {
        const unsigned cand = prefix | (1u << bit);
        const int below = __popcll(__ballot(key < cand));
        prefix = below < K ? cand : prefix;
    }
    const unsigned ub = min(prefix | 0x7fffu, 0xff800000u);      // never past +inf (a NaN pattern would filter everything)
    return __uint_as_float(ub & 0x7fffffffu);
}

__global__ __launch_bounds__(KNN4_THREADS) void knn3_wave4_kernel(
    int n, int m, int K, int qpw, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    int32_t *__restrict__ idx, float *__restrict__ dist2) {
    __shared__ float4 cand[KNN4_TILE];
    __shared__ DI queue[KNN4_WAVES][KNN4_Q][KNN4_QCAP + KNN_FAST_MAX_K];
    __shared__ DI win[KNN4_WAVES][KNN4_Q][KNN_FAST_MAX_K];

    const int bs = blockIdx.y;
    const int lane = lane_id();
    const int wave = threadIdx.x / PDGN_WAVE;
    const float *P = xyz + (size_t)bs * n * 3;
    // qpw groups of KNN4_Q queries per wave, one after the other on the same staged tile (qpw > 1 only when the whole
    // candidate set is one tile: the running lists of a group live in registers across tiles)
  for (int grp = 0; grp < qpw; ++grp) {
    const int q0 = ((blockIdx.x * KNN4_WAVES + wave) * qpw + grp) * KNN4_Q;

    float qx[KNN4_Q], qy[KNN4_Q], qz[KNN4_Q], rd[KNN4_Q];
    int ri[KNN4_Q], cnt[KNN4_Q];
#pragma unroll
    for (int q = 0; q < KNN4_Q; ++q) {
        const int query = min(q0 + q, m - 1);                    // queries past m: computed, never stored
        const float *Qp = new_xyz + ((size_t)bs * m + query) * 3;
        qx[q] = Qp[0]; qy[q] = Qp[1]; qz[q] = Qp[2];
        rd[q] = INFINITY;
        ri[q] = 0x7fffffff;
        cnt[q] = 0;
    }
    const bool vec_ok = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(xyz) & 15) == 0);
    for (int t0 = 0; t0 < n; t0 += KNN4_TILE) {
        const int tn = min(KNN4_TILE, n - t0);
        const int tp = (tn + 255) & ~255;
      if (grp == 0 || n > KNN4_TILE) {
        __syncthreads();                                         // previous tile fully consumed
        if (vec_ok) {
            // 4 points = 12 floats = three float4 per thread (t0 and tn are multiples of 4 here)
            const float4 *src = reinterpret_cast<const float4 *>(P + (size_t)t0 * 3);
            for (int g = threadIdx.x; g < tn / 4; g += KNN4_THREADS) {
                const float4 a = src[3 * g], b = src[3 * g + 1], c = src[3 * g + 2];
                cand[4 * g] = make_float4(a.x, a.y, a.z, 0.f);
                cand[4 * g + 1] = make_float4(a.w, b.x, b.y, 0.f);
                cand[4 * g + 2] = make_float4(b.z, b.w, c.x, 0.f);
                cand[4 * g + 3] = make_float4(c.y, c.z, c.w, 0.f);
            }
        } else {
            for (int c = threadIdx.x; c < tn; c += KNN4_THREADS) {
                const float *pp = P + (size_t)(t0 + c) * 3;
                cand[c] = make_float4(pp[0], pp[1], pp[2], 0.f);
            }
        }
        // pad the tile to a multiple of 256 candidates with points at infinity (distance +inf: never a minimum, never
        // kept), so that the scans below run without per-candidate guards, four LDS reads in flight per lane
        for (int c = tn + threadIdx.x; c < tp; c += KNN4_THREADS) cand[c] = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
        __syncthreads();
      }
        // pass A: per-lane minimum for each of the four queries -> thresholds
        float lmin[KNN4_Q];
#pragma unroll
        for (int q = 0; q < KNN4_Q; ++q) lmin[q] = INFINITY;
        for (int c = lane; c < tp; c += 256) {
            float4 p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = cand[c + 64 * j];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < KNN4_Q; ++q) {
                    const float d = sqdist3(qx[q], qy[q], qz[q], p[j].x, p[j].y, p[j].z);
                    lmin[q] = d < lmin[q] ? d : lmin[q];         // (fminf would add a canonicalising v_max per value)
                }
        }
        float tau[KNN4_Q];
        {   // 16-bit upper bounds of the K-th smallest lane minimum, the queries' bisections interleaved stage by stage
            unsigned key[KNN4_Q], prefix[KNN4_Q];
#pragma unroll
            for (int q = 0; q < KNN4_Q; ++q) { key[q] = __float_as_uint(lmin[q]) | 0x80000000u; prefix[q] = 0x80000000u; }
#pragma unroll
            for (int bit = 30; bit >= 15; --bit) {
                unsigned long long lo[KNN4_Q];
#pragma unroll
                for (int q = 0; q < KNN4_Q; ++q) lo[q] = __ballot(key[q] < (prefix[q] | (1u << bit)));
#pragma unroll
                for (int q = 0; q < KNN4_Q; ++q)
                    prefix[q] = __builtin_amdgcn_readfirstlane(__popcll(lo[q]) < K ? (prefix[q] | (1u << bit)) : prefix[q]);
            }
#pragma unroll
            for (int q = 0; q < KNN4_Q; ++q) {
                const unsigned ub = min(prefix[q] | 0x7fffu, 0xff800000u);
                tau[q] = fminf(__uint_as_float(ub & 0x7fffffffu), __shfl(rd[q], K - 1, 64));
            }
        }
        // pass B: survivors into the per-query queues.  Written for few VALU -> SALU -> VALU round trips: the compares of a
        // step for both queries, ONE scalar overflow test, then the predicated writes at mbcnt slots (wave_select.h).
#pragma unroll
        for (int q = 0; q < KNN4_Q; ++q) tau[q] = fminf(tau[q], 3.402823466e38f);   // finite: "d <= tau" keeps +inf out
        int sc[KNN4_Q];
#pragma unroll
        for (int q = 0; q < KNN4_Q; ++q) sc[q] = __builtin_amdgcn_readfirstlane(cnt[q]);
        for (int c0 = 0; c0 < tp; c0 += 128) {
            float4 p[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) p[j] = cand[c0 + 64 * j + lane];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = c0 + 64 * j + lane;
                float d[KNN4_Q];
                unsigned long long mask[KNN4_Q];
                int add[KNN4_Q];
                bool over = false;
#pragma unroll
                for (int q = 0; q < KNN4_Q; ++q) {
                    d[q] = sqdist3(qx[q], qy[q], qz[q], p[j].x, p[j].y, p[j].z);
                    mask[q] = __ballot(d[q] <= tau[q]);
                }
#pragma unroll
                for (int q = 0; q < KNN4_Q; ++q) {
                    add[q] = __popcll(mask[q]);
                    over = over || sc[q] + add[q] > KNN4_QCAP;
                }
                if (!over) {
#pragma unroll
                    for (int q = 0; q < KNN4_Q; ++q) {
                        const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask[q], sc[q]));
                        if (d[q] <= tau[q]) { queue[wave][q][pos].d = d[q]; queue[wave][q][pos].i = t0 + c; }
                        sc[q] += add[q];
                    }
                } else {                                         // rare (duplicates, > 64 ties): merge, tighten, re-filter
#pragma unroll
                    for (int q = 0; q < KNN4_Q; ++q) {
                        DI *qq = queue[wave][q];
                        bool keep = d[q] <= tau[q];
                        unsigned long long mk = mask[q];
                        if (sc[q] + add[q] > KNN4_QCAP) {
                            knn_flush_ranked(qq, sc[q], K, rd[q], ri[q], lane, win[wave][q]);
                            sc[q] = 0;
                            tau[q] = fminf(tau[q], __shfl(rd[q], K - 1, 64));
                            keep = keep && d[q] <= tau[q];
                            mk = __ballot(keep);
                        }
                        const int pos = sc[q] + __popcll(mk & ((1ull << lane) - 1ull));
                        if (keep) { qq[pos].d = d[q]; qq[pos].i = t0 + c; }
                        sc[q] = __builtin_amdgcn_readfirstlane(sc[q] + __popcll(mk));
                    }
                }
            }
        }
        {
            DI *const qs[KNN4_Q] = {queue[wave][0], queue[wave][1]};
            knn_flush_select_multi<KNN4_Q>(qs, sc, K, rd, ri, lane, win[wave][0]);
        }
#pragma unroll
        for (int q = 0; q < KNN4_Q; ++q) cnt[q] = 0;
    }
#pragma unroll
    for (int q = 0; q < KNN4_Q; ++q) {
        const int query = q0 + q;
        if (query < m && lane < K) {
            const size_t o = ((size_t)bs * m + query) * K + lane;
            const bool valid = rd[q] < INFINITY;
            idx[o] = valid ? ri[q] : 0;                          // reference tail: idx 0 / dist 1e40 -> +inf
            if (dist2) dist2[o] = rd[q];
        }
    }
  }
}

// Generic fallback for 32 < nsample <= 200: one thread per query, sorted insertion in
// per-thread scratch (the reference's own shape; not used by PDGN, which has nsample = 20).
__global__ __launch_bounds__(KNN_THREADS) void knn3_generic_kernel(
    int n, int m, int K, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    int32_t *__restrict__ idx, float *__restrict__ dist2) {
    const int bs = blockIdx.y;
    const int query = blockIdx.x * blockDim.x + threadIdx.x;
    if (query >= m) return;
    const float *P = xyz + (size_t)bs * n * 3;
    const float *Q = new_xyz + ((size_t)bs * m + query) * 3;
    const float qx = Q[0], qy = Q[1], qz = Q[2];
    float best[PDGN_KNN_MAX_NSAMPLE];
    int besti[PDGN_KNN_MAX_NSAMPLE];
    for (int i = 0; i < K; ++i) { best[i] = INFINITY; besti[i] = 0; }
    for (int c = 0; c < n; ++c) {
        float d = sqdist3(qx, qy, qz, P[c * 3], P[c * 3 + 1], P[c * 3 + 2]);
        if (d < best[K - 1]) {
            int j = K - 1;
            while (j > 0 && d < best[j - 1]) { best[j] = best[j - 1]; besti[j] = besti[j - 1]; --j; }
            best[j] = d;
            besti[j] = c;
        }
    }
    size_t o = ((size_t)bs * m + query) * K;
    for (int i = 0; i < K; ++i) {
        idx[o + i] = besti[i];
        if (dist2) dist2[o + i] = best[i];
    }
}

// 3-NN: one thread per unknown point, `known` staged through LDS; three running minima in
// registers with the reference's strict '<' cascade (interpolation_cuda_kernel.cu:157-171).
__global__ __launch_bounds__(KNN_THREADS) void nn3_kernel(
    int n, int m, const float *__restrict__ unknown, const float *__restrict__ known,
    float *__restrict__ dist2, int32_t *__restrict__ idx) {
    __shared__ float4 cand[2048];
    const int bs = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const float *Kn = known + (size_t)bs * m * 3;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (j < n) {
        const float *U = unknown + ((size_t)bs * n + j) * 3;
        ux = U[0]; uy = U[1]; uz = U[2];
    }
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int t0 = 0; t0 < m; t0 += 2048) {
        const int tn = min(2048, m - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < tn * 3; e += KNN_THREADS)
            reinterpret_cast<float *>(cand)[(e / 3) * 4 + (e % 3)] = Kn[(size_t)t0 * 3 + e];
        __syncthreads();
        for (int c = 0; c < tn; ++c) {
            float4 p = cand[c];
            float d = sqdist3(ux, uy, uz, p.x, p.y, p.z);
            int k = t0 + c;
            if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
            else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
            else if (d < b3) { b3 = d; i3 = k; }
        }
    }
    if (j < n) {
        size_t o = ((size_t)bs * n + j) * 3;
        dist2[o] = b1; dist2[o + 1] = b2; dist2[o + 2] = b3;
        idx[o] = i1; idx[o + 1] = i2; idx[o + 2] = i3;
    }
}

extern "C" int pdgn_knnquery(int b, int n, int m, int nsample, const float *xyz,
                             const float *new_xyz, int32_t *idx, float *dist2,
                             pdgn_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || nsample < 1 || nsample > PDGN_KNN_MAX_NSAMPLE) return PDGN_ERR_INVALID;
    if (b == 0 || m == 0) return 0;
    if (b > 65535) return PDGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (nsample <= KNN_FAST_MAX_K) {
        // one query group per wave: measured fastest at the step's shapes (tools/knn3_bench.py)
        dim3 grid(cdiv(m, KNN4_WAVES * KNN4_Q), b);
        hipLaunchKernelGGL(knn3_wave4_kernel, grid, dim3(KNN4_THREADS), 0, s, n, m, nsample, 1, xyz, new_xyz, idx, dist2);
    } else {
        dim3 grid(cdiv(m, KNN_THREADS), b);
        hipLaunchKernelGGL(knn3_generic_kernel, grid, dim3(KNN_THREADS), 0, s, n, m, nsample, xyz,
                           new_xyz, idx, dist2);
    }
    return pdgn_launch_status();
}

extern "C" int pdgn_nearestneighbor(int b, int n, int m, const float *unknown, const float *known,
                                    float *dist2, int32_t *idx, pdgn_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0 || n == 0) return 0;
    dim3 grid(cdiv(n, KNN_THREADS), b);
    hipLaunchKernelGGL(nn3_kernel, grid, dim3(KNN_THREADS), 0, (hipStream_t)stream, n, m, unknown,
                       known, dist2, idx);
    return pdgn_launch_status();
}
