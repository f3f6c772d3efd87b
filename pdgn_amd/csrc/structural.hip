// structural.hip -- Chamfer nearest-neighbour distance and approximate EMD for gfx950.
//
// Semantics: evaluation/pytorch_structural_losses/src/nndistance.cu:2-154 and
// src/approxmatch.cu:3-326 of the reference.
//
//  * nndistance: one thread per point, the other cloud staged through LDS as float4 tiles,
//    both directions in ONE launch (blockIdx.z).  The reference launches a fixed (32,16)
//    grid twice; here the grid covers (points/256, batch, 2).
//  * approxmatch: ONE WORKGROUP PER PAIR (the reference pins the grid at 32 blocks and
//    serialises b/32 pairs per block).  Each thread owns PPT points, so one broadcast
//    ds_read_b128 of the opposite cloud feeds PPT exp evaluations.  remain/ratio vectors
//    stay in the caller's `temp` scratch (a few KB per pair, L2-resident).
//  * emd_cost: the same 9-level auction with phase 3 accumulating match*dist on the fly --
//    the (b,m,n) match matrix (8.6 GB at b=512, 2048^2) never touches HBM.
#include "common.h"
#include <type_traits>

#define SL_THREADS 256
#define SL_TILE 2048

// ---------------------------------------------------------------------------- nn distance
// nndistance.cu:21-24: d = x2*x2+y2*y2+z2*z2 with x2 = cand - query, contracted by nvcc to
// fma(z2,z2, fma(y2,y2, x2*x2)); strict '<' => lowest index wins ties (:26, :116).
__global__ __launch_bounds__(SL_THREADS) void nndist_kernel(
    int n, int m, const float *__restrict__ xyz, const float *__restrict__ xyz2,
    float *__restrict__ res, int32_t *__restrict__ res_i, float *__restrict__ res2,
    int32_t *__restrict__ res2_i) {
    __shared__ float4 cand[SL_TILE];
    const int bs = blockIdx.y;
    const bool rev = blockIdx.z != 0;
    const int nq = rev ? m : n, nc = rev ? n : m;
    const float *Q = (rev ? xyz2 : xyz) + (size_t)bs * nq * 3;
    const float *C = (rev ? xyz : xyz2) + (size_t)bs * nc * 3;
    float *out = (rev ? res2 : res) + (size_t)bs * nq;
    int32_t *out_i = (rev ? res2_i : res_i) + (size_t)bs * nq;
    if ((int)(blockIdx.x * blockDim.x) >= nq) return;   // block-uniform
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (j < nq) { qx = Q[j * 3]; qy = Q[j * 3 + 1]; qz = Q[j * 3 + 2]; }
    float best = 0.f;
    int best_i = 0;
    for (int t0 = 0; t0 < nc; t0 += SL_TILE) {
        const int tn = min(SL_TILE, nc - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < tn * 3; e += SL_THREADS)
            reinterpret_cast<float *>(cand)[(e / 3) * 4 + (e % 3)] = C[(size_t)t0 * 3 + e];
        __syncthreads();
        for (int c = 0; c < tn; ++c) {
            float4 p = cand[c];
            float x2 = p.x - qx, y2 = p.y - qy, z2 = p.z - qz;
            float d = __fmaf_rn(z2, z2, __fmaf_rn(y2, y2, __fmul_rn(x2, x2)));
            bool better = (t0 + c == 0) || d < best;
            best = better ? d : best;
            best_i = better ? t0 + c : best_i;
        }
    }
    if (j < nq) { out[j] = best; out_i[j] = best_i; }
}

// nndistance.cu:129-148: g = 2*grad; grad1[j] += g*(p1-p2); grad2[idx] -= g*(p1-p2).
__global__ __launch_bounds__(SL_THREADS) void nndist_grad_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ gd1, const int32_t *__restrict__ idx1, const float *__restrict__ gd2,
    const int32_t *__restrict__ idx2, float *__restrict__ g1, float *__restrict__ g2) {
    const int bs = blockIdx.y;
    const bool rev = blockIdx.z != 0;
    const int na = rev ? m : n, nb = rev ? n : m;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= na) return;
    const float *A = (rev ? xyz2 : xyz1) + (size_t)bs * na * 3;
    const float *B = (rev ? xyz1 : xyz2) + (size_t)bs * nb * 3;
    const float *gd = (rev ? gd2 : gd1) + (size_t)bs * na;
    const int32_t *id = (rev ? idx2 : idx1) + (size_t)bs * na;
    float *ga = (rev ? g2 : g1) + (size_t)bs * na * 3;
    float *gb = (rev ? g1 : g2) + (size_t)bs * nb * 3;
    const int j2 = id[j];
    const float g = gd[j] * 2;
    for (int c = 0; c < 3; ++c) {
        float v = g * (A[j * 3 + c] - B[j2 * 3 + c]);
        atomicAdd(&ga[j * 3 + c], v);
        atomicAdd(&gb[j2 * 3 + c], -v);
    }
}

// ---------------------------------------------------------------------------- approx EMD
#define AM_THREADS 512
#define AM_PPT 4                      // points per thread per sweep
#define AM_SWEEP (AM_THREADS * AM_PPT)
#define AM_TILE 1024                  // opposite-cloud tile (approxmatch.cu:13 Block=1024)

__device__ __forceinline__ float sq3(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

// Packed-fp32 forms of the inner loops: two points of a thread ride in one VGPR pair, so the subtract / multiply /
// fused-multiply-add chain of every (point, opposite point) element issues as v_pk_* instructions at half the
// per-element cost; only the transcendentals (v_exp_f32, v_sqrt_f32: quarter rate) stay scalar.  Same operations in the
// same order as sq3 / __expf, so the results are bit-identical to the scalar form.
typedef float f2 __attribute__((ext_vector_type(2)));
#define AM_LOG2E 0x1.715476p+0f
__device__ __forceinline__ f2 sq3_pk(float qx, float qy, float qz, f2 px, f2 py, f2 pz, bool q_first) {
    const f2 vx = {qx, qx}, vy = {qy, qy}, vz = {qz, qz};
    const f2 dx = q_first ? vx - px : px - vx, dy = q_first ? vy - py : py - vy, dz = q_first ? vz - pz : pz - vz;
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}
__device__ __forceinline__ f2 exp_pk(float level, f2 d2) {              // __expf(level * d2) per lane
    const f2 lv = {level, level}, k = {AM_LOG2E, AM_LOG2E};
    const f2 a = (lv * d2) * k;
    f2 r;
    r.x = __builtin_amdgcn_exp2f(a.x);
    r.y = __builtin_amdgcn_exp2f(a.y);
    return r;
}

// Writes match (b,m,n) like approxmatchkernel (approxmatch.cu:3-182), line by line: phases 1 and 2 with packed arithmetic,
// phase 3 is the read-modify-write of `match` (HBM-bound).  The cost-only evaluation path uses emd_cost_kernel below.
__global__ __launch_bounds__(AM_THREADS) void approxmatch_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    float *__restrict__ match, float *__restrict__ temp) {
    __shared__ float4 buf[AM_TILE];
    const int pair = blockIdx.x;
    const float *A = xyz1 + (size_t)pair * n * 3;
    const float *B = xyz2 + (size_t)pair * m * 3;
    float *M = match + (size_t)pair * n * m;
    // remainL | remainR | ratioL | ratioR, as approxmatch.cu:4 lays out `temp`
    float *remainL = temp + (size_t)pair * (n + m) * 2, *remainR = remainL + n,
          *ratioL = remainR + m, *ratioR = ratioL + n;
    const float multiL = n >= m ? 1.f : (float)(m / n);     // integer division, :6-12
    const float multiR = n >= m ? (float)(n / m) : 1.f;
    const int tid = threadIdx.x;
    for (int k = tid; k < n; k += AM_THREADS) remainL[k] = multiL;
    for (int l = tid; l < m; l += AM_THREADS) remainR[l] = multiR;
    __syncthreads();

    for (int j = 7; j > -2; --j) {                          // 9 levels; the j==-2 branch is dead
        const float level = -ldexpf(1.0f, 2 * j);           // -4^j
        // ---- phase 1 (:29-62): ratioL[k] = remainL[k] / (1e-9 + sum_l exp(level*d2)*remainR[l])
        for (int k0 = 0; k0 < n; k0 += AM_SWEEP) {
            float px[AM_PPT], py[AM_PPT], pz[AM_PPT], acc[AM_PPT];
#pragma unroll
            for (int i = 0; i < AM_PPT; ++i) {
                int k = k0 + tid + i * AM_THREADS;
                bool ok = k < n;
                px[i] = ok ? A[k * 3] : 0.f; py[i] = ok ? A[k * 3 + 1] : 0.f; pz[i] = ok ? A[k * 3 + 2] : 0.f;
                acc[i] = 1e-9f;
            }
            for (int l0 = 0; l0 < m; l0 += AM_TILE) {
                const int lend = min(AM_TILE, m - l0);
                __syncthreads();
                for (int l = tid; l < lend; l += AM_THREADS)
                    buf[l] = make_float4(B[(l0 + l) * 3], B[(l0 + l) * 3 + 1], B[(l0 + l) * 3 + 2], remainR[l0 + l]);
                __syncthreads();
                f2 ax[AM_PPT / 2], ay[AM_PPT / 2], az[AM_PPT / 2], ac[AM_PPT / 2];
#pragma unroll
                for (int i = 0; i < AM_PPT / 2; ++i) {
                    ax[i] = f2{px[2 * i], px[2 * i + 1]}; ay[i] = f2{py[2 * i], py[2 * i + 1]};
                    az[i] = f2{pz[2 * i], pz[2 * i + 1]}; ac[i] = f2{acc[2 * i], acc[2 * i + 1]};
                }
                for (int l = 0; l < lend; ++l) {
                    const float4 q = buf[l];
                    const f2 qw = {q.w, q.w};
#pragma unroll
                    for (int i = 0; i < AM_PPT / 2; ++i)
                        ac[i] = __builtin_elementwise_fma(exp_pk(level, sq3_pk(q.x, q.y, q.z, ax[i], ay[i], az[i], true)), qw, ac[i]);
                }
#pragma unroll
                for (int i = 0; i < AM_PPT / 2; ++i) { acc[2 * i] = ac[i].x; acc[2 * i + 1] = ac[i].y; }
            }
#pragma unroll
            for (int i = 0; i < AM_PPT; ++i) {
                int k = k0 + tid + i * AM_THREADS;
                if (k < n) ratioL[k] = remainL[k] / acc[i];
            }
        }
        __syncthreads();
        // ---- phase 2 (:78-111)
        for (int l0 = 0; l0 < m; l0 += AM_SWEEP) {
            float px[AM_PPT], py[AM_PPT], pz[AM_PPT], acc[AM_PPT];
#pragma unroll
            for (int i = 0; i < AM_PPT; ++i) {
                int l = l0 + tid + i * AM_THREADS;
                bool ok = l < m;
                px[i] = ok ? B[l * 3] : 0.f; py[i] = ok ? B[l * 3 + 1] : 0.f; pz[i] = ok ? B[l * 3 + 2] : 0.f;
                acc[i] = 0.f;
            }
            for (int k0 = 0; k0 < n; k0 += AM_TILE) {
                const int kend = min(AM_TILE, n - k0);
                __syncthreads();
                for (int k = tid; k < kend; k += AM_THREADS)
                    buf[k] = make_float4(A[(k0 + k) * 3], A[(k0 + k) * 3 + 1], A[(k0 + k) * 3 + 2], ratioL[k0 + k]);
                __syncthreads();
                f2 ax[AM_PPT / 2], ay[AM_PPT / 2], az[AM_PPT / 2], ac[AM_PPT / 2];
#pragma unroll
                for (int i = 0; i < AM_PPT / 2; ++i) {
                    ax[i] = f2{px[2 * i], px[2 * i + 1]}; ay[i] = f2{py[2 * i], py[2 * i + 1]};
                    az[i] = f2{pz[2 * i], pz[2 * i + 1]}; ac[i] = f2{acc[2 * i], acc[2 * i + 1]};
                }
                for (int k = 0; k < kend; ++k) {
                    const float4 q = buf[k];
                    const f2 qw = {q.w, q.w};
#pragma unroll
                    for (int i = 0; i < AM_PPT / 2; ++i)
                        ac[i] = __builtin_elementwise_fma(exp_pk(level, sq3_pk(q.x, q.y, q.z, ax[i], ay[i], az[i], false)), qw, ac[i]);
                }
#pragma unroll
                for (int i = 0; i < AM_PPT / 2; ++i) { acc[2 * i] = ac[i].x; acc[2 * i + 1] = ac[i].y; }
            }
#pragma unroll
            for (int i = 0; i < AM_PPT; ++i) {
                int l = l0 + tid + i * AM_THREADS;
                if (l < m) {
                    float r = remainR[l];
                    float sumr = acc[i] * r;
                    float consumption = fminf(r / (sumr + 1e-9f), 1.0f);
                    ratioR[l] = consumption * r;
                    remainR[l] = fmaxf(0.0f, r - sumr);
                }
            }
        }
        __syncthreads();
        // ---- phase 3 (:130-163): w = exp(.)*ratioL[k]*ratioR[l]; match[l,k] += w; remainL -= sum_l w
        for (int k0 = 0; k0 < n; k0 += AM_SWEEP) {
            float px[AM_PPT], py[AM_PPT], pz[AM_PPT], rl[AM_PPT], acc[AM_PPT];
#pragma unroll
            for (int i = 0; i < AM_PPT; ++i) {
                int k = k0 + tid + i * AM_THREADS;
                bool ok = k < n;
                px[i] = ok ? A[k * 3] : 0.f; py[i] = ok ? A[k * 3 + 1] : 0.f; pz[i] = ok ? A[k * 3 + 2] : 0.f;
                rl[i] = ok ? ratioL[k] : 0.f;             // the reference reads ratioL[k>=n] OOB (:148)
                acc[i] = 0.f;
            }
            for (int l0 = 0; l0 < m; l0 += AM_TILE) {
                const int lend = min(AM_TILE, m - l0);
                __syncthreads();
                for (int l = tid; l < lend; l += AM_THREADS)
                    buf[l] = make_float4(B[(l0 + l) * 3], B[(l0 + l) * 3 + 1], B[(l0 + l) * 3 + 2], ratioR[l0 + l]);
                __syncthreads();
                for (int l = 0; l < lend; ++l) {
                    float4 q = buf[l];
#pragma unroll
                    for (int i = 0; i < AM_PPT; ++i) {
                        int k = k0 + tid + i * AM_THREADS;
                        float d2 = sq3(q.x, q.y, q.z, px[i], py[i], pz[i]);
                        float w = __expf(level * d2) * rl[i] * q.w;
                        if (k < n) {
                            float *dst = &M[(size_t)(l0 + l) * n + k];
                            *dst = (j == 7) ? w : *dst + w;   // first level overwrites: no zero-fill pass
                        }
                        acc[i] += w;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < AM_PPT; ++i) {
                int k = k0 + tid + i * AM_THREADS;
                if (k < n) remainL[k] = fmaxf(0.0f, remainL[k] - acc[i]);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------- fused EMD cost (round 3)
// The same 9-level auction as approxmatch_kernel, restructured around what its inner loops cost on gfx950 (one
// v_exp_f32 = two full-rate slots; every other operation one packed slot per two points):
//  * the exponent's two multiplies are gone: coordinates are pre-scaled per level by s_j = 2^j sqrt(log2 e), so that the
//    squared distance of the scaled points IS the (negated) v_exp_f32 argument, 4^j |p - q|^2 log2 e;
//  * phase 3 of level j and phase 1 of level j-1 sweep the same (k, all l) pairs and depend on each other only through
//    remainL[k], which phase 1 needs as a numerator AFTER its sum: they run as ONE sweep.  Its exponential is taken at
//    the lower level, e' = exp(-4^(j-1) d2), and the upper level's is e'^4 (two packed multiplies), so a level costs two
//    sweeps and two exponentials per pair of points instead of three and three: 19 n m exponentials per cloud pair, not 27;
//  * sqrt(d2) of the cost term is v_sqrt_f32 of the scaled squared distance, un-scaled once per sweep.
// Same phase order, guards (1e-9), integer multiL / multiR and clamps as approxmatch.cu:3-224; results agree with the
// line-by-line form to ~1e-6 relative (the tests hold it to 1e-4 against the C oracle).
#define EMD_SQRT_LOG2E 0x1.337f14p+0f          // sqrt(log2 e) = 1.2011224
__device__ __forceinline__ f2 exp2neg_pk(f2 a) { return f2{__builtin_amdgcn_exp2f(-a.x), __builtin_amdgcn_exp2f(-a.y)}; }

// Sorted sweeps (round 3, second step).  exp(-4^j |p - q|^2) is EXACTLY zero in fp32 once 4^j log2e |p - q|^2 > 150, and
// |p - q| >= |p.x - q.x|: with both clouds sorted by x (a bitonic sort in LDS at kernel start; the cost is invariant under
// permutations of either cloud) a wave whose 256 own points span [xlo, xhi] needs, of a staged tile, only the contiguous run of
// points with x in [xlo - r_j, xhi + r_j], r_j = sqrt(150 / (4^j log2 e)) -- two binary searches per tile and wave, no per-element
// test.  Everything outside the run would have contributed exact zeros.  At the four sharpest levels that is 20-45 % of a tile
// for uniform clouds (r_7 = 0.08 against a cloud width of 2); from level 3 down r_j exceeds the cloud and nothing is skipped.
// SORTED = false: the clouds as they are, whole tiles (n or m > 2048: the sort's LDS image holds 2048 points).
#define EMD_ZERO_ARG 12.2475f                   // sqrt(150): scaled |dx| beyond which exp2(-d2) is zero, denormals included
template <bool SORTED>
__global__ __launch_bounds__(AM_THREADS) void emd_cost_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2, float *__restrict__ temp,
    float *__restrict__ cost_out, const int32_t *__restrict__ ia, const int32_t *__restrict__ ib) {
    __shared__ float4 sbuf[2048];              // the sort's image; afterwards buf (staged tile) | buf2 (second weight)
    __shared__ float red[AM_THREADS / PDGN_WAVE];
    float4 *buf = sbuf;                        // scaled opposite point + its weight of the phase being evaluated
    float *buf2 = reinterpret_cast<float *>(sbuf + AM_TILE);   // merged sweeps: remainR[l] (the weight of the NEXT level's phase 1)
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *A0 = xyz1 + (size_t)(ia ? ia[pair] : pair) * n * 3;
    const float *B0 = xyz2 + (size_t)(ib ? ib[pair] : pair) * m * 3;
    // scratch of a pair: remainL | remainR | ratioL | ratioR (as approxmatch.cu:4), then the two clouds as float4 rows
    const size_t head = ((size_t)(n + m) * 2 + 3) & ~(size_t)3;
    float *base = temp + (size_t)pair * (head + (size_t)(n + m) * 4);
    float *remainL = base, *remainR = remainL + n, *ratioL = remainR + m, *ratioR = ratioL + n;
    float4 *A = reinterpret_cast<float4 *>(base + head), *B = A + n;
    const float multiL = n >= m ? 1.f : (float)(m / n);     // integer division, approxmatch.cu:6-12
    const float multiR = n >= m ? (float)(n / m) : 1.f;
    for (int k = tid; k < n; k += AM_THREADS) remainL[k] = multiL;
    for (int l = tid; l < m; l += AM_THREADS) remainR[l] = multiR;
    auto stage_cloud = [&](const float *P, int cnt, float4 *out) {
        if (SORTED) {
            int p2 = 1;
            while (p2 < cnt) p2 <<= 1;
            for (int i = tid; i < p2; i += AM_THREADS)
                sbuf[i] = i < cnt ? make_float4(P[i * 3], P[i * 3 + 1], P[i * 3 + 2], 0.f) : make_float4(INFINITY, 0.f, 0.f, 0.f);
            __syncthreads();
            for (int kk = 2; kk <= p2; kk <<= 1)
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < p2 / 2; t += AM_THREADS) {
                        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
                        const float4 u = sbuf[i], v = sbuf[ixj];
                        if ((u.x > v.x) == ((i & kk) == 0)) { sbuf[i] = v; sbuf[ixj] = u; }
                    }
                    __syncthreads();
                }
            for (int i = tid; i < cnt; i += AM_THREADS) out[i] = sbuf[i];
        } else {
            for (int i = tid; i < cnt; i += AM_THREADS) out[i] = make_float4(P[i * 3], P[i * 3 + 1], P[i * 3 + 2], 0.f);
        }
        __syncthreads();
    };
    stage_cloud(A0, n, A);
    stage_cloud(B0, m, B);
    float cost = 0.f;

    // [lo, hi) of the staged tile (x ascending, scaled by s) that can contribute to own points with scaled x in [xlo, xhi]
    auto active_run = [&](int cnt, float xlo, float xhi, int &lo, int &hi) {
        if (!SORTED) { lo = 0; hi = cnt; return; }
        const float a = xlo - EMD_ZERO_ARG, b = xhi + EMD_ZERO_ARG;
        int l0 = 0, l1 = cnt;
        while (l0 < l1) { const int mid = (l0 + l1) >> 1; if (buf[mid].x < a) l0 = mid + 1; else l1 = mid; }
        lo = l0;
        l1 = cnt;
        while (l0 < l1) { const int mid = (l0 + l1) >> 1; if (buf[mid].x <= b) l0 = mid + 1; else l1 = mid; }
        hi = l0;
    };

    // one sweep with the thread owning points of cloud A: MODE 0 = phase 1 of level j (first level only),
    // 1 = phase 3 of level j merged with phase 1 of level j-1 (coordinates scaled for level j-1), 2 = phase 3 only (last level)
    auto sweep_a = [&](auto mode_c, const float s) {
        constexpr int mode = decltype(mode_c)::value;
        for (int k0 = 0; k0 < n; k0 += AM_SWEEP) {
            f2 ax[AM_PPT / 2], ay[AM_PPT / 2], az[AM_PPT / 2], ar[AM_PPT / 2], a1[AM_PPT / 2], a3[AM_PPT / 2], cs[AM_PPT / 2];
            const int kw = k0 + wave * (64 * AM_PPT);           // a wave's own points are a contiguous (x-sorted) run
#pragma unroll
            for (int i = 0; i < AM_PPT / 2; ++i) {
                float v[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = kw + (2 * i + h) * 64 + lane;
                    const bool ok = k < n;
                    const float4 pt = ok ? A[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                    v[h][0] = pt.x * s; v[h][1] = pt.y * s; v[h][2] = pt.z * s;
                    v[h][3] = (ok && mode != 0) ? ratioL[k] : 0.f;
                }
                ax[i] = f2{v[0][0], v[1][0]}; ay[i] = f2{v[0][1], v[1][1]}; az[i] = f2{v[0][2], v[1][2]};
                ar[i] = f2{v[0][3], v[1][3]};
                a1[i] = f2{1e-9f, 1e-9f}; a3[i] = f2{0.f, 0.f}; cs[i] = f2{0.f, 0.f};
            }
            const bool wlive = kw < n;
            const float xlo = wlive ? A[kw].x * s : 0.f, xhi = wlive ? A[min(n, kw + 64 * AM_PPT) - 1].x * s : 0.f;
            for (int l0 = 0; l0 < m; l0 += AM_TILE) {
                const int lend = min(AM_TILE, m - l0);
                __syncthreads();
                const int lpad = (lend + 3) & ~3;           // the loops below take four staged points per trip: the tail is
                for (int l = tid; l < lpad; l += AM_THREADS) {       // padded with weight-zero points (they add exact zeros)
                    const bool ok = l < lend;
                    const float w0 = !ok ? 0.f : (mode == 0 ? remainR[l0 + l] : ratioR[l0 + l]);
                    const float4 q = ok ? B[l0 + l] : make_float4(0.f, 0.f, 0.f, 0.f);
                    buf[l] = make_float4(q.x * s, q.y * s, q.z * s, w0);
                    if (mode == 1) buf2[l] = ok ? remainR[l0 + l] : 0.f;
                }
                __syncthreads();
                int lo, hi;
                active_run(lend, xlo, xhi, lo, hi);
                if (!wlive) hi = lo = 0;
                const int lb = lo & ~3, le = min(lpad, (hi + 3) & ~3);
                if (mode == 0) {
#pragma unroll 4
                    for (int l = lb; l < le; ++l) {
                        const float4 q = buf[l];
                        const f2 qw = {q.w, q.w};
#pragma unroll
                        for (int i = 0; i < AM_PPT / 2; ++i)
                            a1[i] = __builtin_elementwise_fma(exp2neg_pk(sq3_pk(q.x, q.y, q.z, ax[i], ay[i], az[i], true)), qw, a1[i]);
                    }
                } else if (mode == 1) {
#pragma unroll 4
                    for (int l = lb; l < le; ++l) {
                        const float4 q = buf[l];
                        const float rr = buf2[l];
                        const f2 qw = {q.w, q.w}, qr = {rr, rr};
#pragma unroll
                        for (int i = 0; i < AM_PPT / 2; ++i) {
                            const f2 d2 = sq3_pk(q.x, q.y, q.z, ax[i], ay[i], az[i], true);
                            const f2 e1 = exp2neg_pk(d2);                    // exp(-4^(j-1) |p-q|^2): next level's phase 1
                            const f2 e2 = e1 * e1;
                            const f2 w = ((e2 * e2) * ar[i]) * qw;           // exp(-4^j |p-q|^2) ratioL[k] ratioR[l]
                            const f2 r = {__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
                            cs[i] = __builtin_elementwise_fma(w, r, cs[i]);
                            a3[i] += w;
                            a1[i] = __builtin_elementwise_fma(e1, qr, a1[i]);
                        }
                    }
                } else {
#pragma unroll 4
                    for (int l = lb; l < le; ++l) {
                        const float4 q = buf[l];
                        const f2 qw = {q.w, q.w};
#pragma unroll
                        for (int i = 0; i < AM_PPT / 2; ++i) {
                            const f2 d2 = sq3_pk(q.x, q.y, q.z, ax[i], ay[i], az[i], true);
                            const f2 w = (exp2neg_pk(d2) * ar[i]) * qw;
                            const f2 r = {__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
                            cs[i] = __builtin_elementwise_fma(w, r, cs[i]);
                            a3[i] += w;
                        }
                    }
                }
            }
            float csum = 0.f;
#pragma unroll
            for (int i = 0; i < AM_PPT / 2; ++i) {
                csum += cs[i].x + cs[i].y;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = kw + (2 * i + h) * 64 + lane;
                    if (k < n) {
                        float rl = remainL[k];
                        if (mode != 0) {
                            rl = fmaxf(0.0f, rl - (h ? a3[i].y : a3[i].x));   // :160-162
                            remainL[k] = rl;
                        }
                        if (mode != 2) ratioL[k] = rl / (h ? a1[i].y : a1[i].x);   // :59-61 (of the next level when merged)
                    }
                }
            }
            cost += csum / s;                                                // sqrt of the scaled d2 = s * distance
        }
    };

    sweep_a(std::integral_constant<int, 0>{}, ldexpf(EMD_SQRT_LOG2E, 7));
    for (int j = 7; j > -2; --j) {                          // 9 levels; the reference's j == -2 branch is dead
        const float s = ldexpf(EMD_SQRT_LOG2E, j);          // s^2 = 4^j log2 e
        __syncthreads();
        // ---- phase 2 (:78-111): the thread owns points of cloud B
        for (int l0 = 0; l0 < m; l0 += AM_SWEEP) {
            f2 ax[AM_PPT / 2], ay[AM_PPT / 2], az[AM_PPT / 2], ac[AM_PPT / 2];
            const int lw = l0 + wave * (64 * AM_PPT);
#pragma unroll
            for (int i = 0; i < AM_PPT / 2; ++i) {
                float v[2][3];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int l = lw + (2 * i + h) * 64 + lane;
                    const float4 pt = l < m ? B[l] : make_float4(0.f, 0.f, 0.f, 0.f);
                    v[h][0] = pt.x * s; v[h][1] = pt.y * s; v[h][2] = pt.z * s;
                }
                ax[i] = f2{v[0][0], v[1][0]}; ay[i] = f2{v[0][1], v[1][1]}; az[i] = f2{v[0][2], v[1][2]};
                ac[i] = f2{0.f, 0.f};
            }
            const bool wlive = lw < m;
            const float xlo = wlive ? B[lw].x * s : 0.f, xhi = wlive ? B[min(m, lw + 64 * AM_PPT) - 1].x * s : 0.f;
            for (int k0 = 0; k0 < n; k0 += AM_TILE) {
                const int kend = min(AM_TILE, n - k0);
                __syncthreads();
                const int kpad = (kend + 3) & ~3;
                for (int k = tid; k < kpad; k += AM_THREADS) {
                    const bool ok = k < kend;
                    const float4 q = ok ? A[k0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
                    buf[k] = make_float4(q.x * s, q.y * s, q.z * s, ok ? ratioL[k0 + k] : 0.f);
                }
                __syncthreads();
                int lo, hi;
                active_run(kend, xlo, xhi, lo, hi);
                if (!wlive) hi = lo = 0;
                const int kb = lo & ~3, ke = min(kpad, (hi + 3) & ~3);
#pragma unroll 4
                for (int k = kb; k < ke; ++k) {
                    const float4 q = buf[k];
                    const f2 qw = {q.w, q.w};
#pragma unroll
                    for (int i = 0; i < AM_PPT / 2; ++i)
                        ac[i] = __builtin_elementwise_fma(exp2neg_pk(sq3_pk(q.x, q.y, q.z, ax[i], ay[i], az[i], false)), qw, ac[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < AM_PPT / 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int l = lw + (2 * i + h) * 64 + lane;
                    if (l < m) {
                        const float r = remainR[l];
                        const float sumr = (h ? ac[i].y : ac[i].x) * r;
                        const float consumption = fminf(r / (sumr + 1e-9f), 1.0f);
                        ratioR[l] = consumption * r;
                        remainR[l] = fmaxf(0.0f, r - sumr);
                    }
                }
        }
        __syncthreads();
        // ---- phase 3 of this level (:130-163) [+ phase 1 of the next one (:29-62)]
        if (j > -1) sweep_a(std::integral_constant<int, 1>{}, ldexpf(EMD_SQRT_LOG2E, j - 1));
        else sweep_a(std::integral_constant<int, 2>{}, s);
    }
    for (int off = 32; off > 0; off >>= 1) cost += __shfl_down(cost, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = cost;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < AM_THREADS / PDGN_WAVE; ++w) t += red[w];
        cost_out[pair] = t;
    }
}

// matchcostkernel approxmatch.cu:184-224: out[b] = sum_{k<m, j<n} match[k*n+j]*sqrt(|xyz2[k]-xyz1[j]|^2)
__global__ __launch_bounds__(AM_THREADS) void matchcost_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ match, float *__restrict__ out) {
    __shared__ float4 buf[AM_TILE];
    __shared__ float red[AM_THREADS / PDGN_WAVE];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const float *A = xyz1 + (size_t)pair * n * 3;
    const float *B = xyz2 + (size_t)pair * m * 3;
    const float *M = match + (size_t)pair * n * m;
    float sub = 0.f;
    for (int k0 = 0; k0 < m; k0 += AM_TILE) {
        const int kend = min(AM_TILE, m - k0);
        __syncthreads();
        for (int k = tid; k < kend; k += AM_THREADS)
            buf[k] = make_float4(B[(k0 + k) * 3], B[(k0 + k) * 3 + 1], B[(k0 + k) * 3 + 2], 0.f);
        __syncthreads();
        for (int j = tid; j < n; j += AM_THREADS) {
            const float x1 = A[j * 3], y1 = A[j * 3 + 1], z1 = A[j * 3 + 2];
            for (int k = 0; k < kend; ++k) {
                float4 q = buf[k];
                sub = __fmaf_rn(M[(size_t)(k0 + k) * n + j], sqrtf(sq3(q.x, q.y, q.z, x1, y1, z1)), sub);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) sub += __shfl_down(sub, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = sub;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < AM_THREADS / PDGN_WAVE; ++w) s += red[w];
        out[pair] = s;
    }
}

// matchcostgrad1kernel approxmatch.cu:270-291: grad1[l] = sum_k (p1-p2)*match[k,l]*rsqrt(max(d2,1e-20))
__global__ __launch_bounds__(SL_THREADS) void matchcost_grad1_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ match, float *__restrict__ grad1) {
    __shared__ float4 buf[AM_TILE];
    const int pair = blockIdx.y, tid = threadIdx.x;
    const int l = blockIdx.x * blockDim.x + tid;
    const float *A = xyz1 + (size_t)pair * n * 3;
    const float *B = xyz2 + (size_t)pair * m * 3;
    const float *M = match + (size_t)pair * n * m;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (l < n) { x1 = A[l * 3]; y1 = A[l * 3 + 1]; z1 = A[l * 3 + 2]; }
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int k0 = 0; k0 < m; k0 += AM_TILE) {
        const int kend = min(AM_TILE, m - k0);
        __syncthreads();
        for (int k = tid; k < kend; k += SL_THREADS)
            buf[k] = make_float4(B[(k0 + k) * 3], B[(k0 + k) * 3 + 1], B[(k0 + k) * 3 + 2], 0.f);
        __syncthreads();
        if (l < n)
            for (int k = 0; k < kend; ++k) {
                float4 q = buf[k];
                float d = M[(size_t)(k0 + k) * n + l] * rsqrtf(fmaxf(sq3(x1, y1, z1, q.x, q.y, q.z), 1e-20f));
                gx += (x1 - q.x) * d; gy += (y1 - q.y) * d; gz += (z1 - q.z) * d;
            }
    }
    if (l < n) {
        float *g = grad1 + ((size_t)pair * n + l) * 3;
        g[0] = gx; g[1] = gy; g[2] = gz;
    }
}

// matchcostgrad2kernel approxmatch.cu:229-269: grad2[k] = sum_j (p2-p1)*match[k,j]*rsqrt(...);
// one wave per xyz2 point, lanes stride the (contiguous) match row.
__global__ __launch_bounds__(SL_THREADS) void matchcost_grad2_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ match, float *__restrict__ grad2) {
    const int pair = blockIdx.y;
    const int k = blockIdx.x * (SL_THREADS / PDGN_WAVE) + threadIdx.x / PDGN_WAVE;
    if (k >= m) return;
    const int lane = lane_id();
    const float *A = xyz1 + (size_t)pair * n * 3;
    const float *B = xyz2 + ((size_t)pair * m + k) * 3;
    const float *M = match + ((size_t)pair * m + k) * n;
    const float x2 = B[0], y2 = B[1], z2 = B[2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int j = lane; j < n; j += 64) {
        float x1 = x2 - A[j * 3], y1 = y2 - A[j * 3 + 1], z1 = z2 - A[j * 3 + 2];
        float d = M[j] * rsqrtf(fmaxf(__fmaf_rn(z1, z1, __fmaf_rn(y1, y1, x1 * x1)), 1e-20f));
        gx += x1 * d; gy += y1 * d; gz += z1 * d;
    }
    for (int off = 32; off > 0; off >>= 1) {
        gx += __shfl_down(gx, off, 64); gy += __shfl_down(gy, off, 64); gz += __shfl_down(gz, off, 64);
    }
    if (lane == 0) {
        float *g = grad2 + ((size_t)pair * m + k) * 3;
        g[0] = gx; g[1] = gy; g[2] = gz;
    }
}

// ---------------------------------------------------------------------------- C ABI
static bool sl_dims_ok(int b, int n, int m) { return b >= 0 && n >= 0 && m >= 0 && b <= 65535; }

extern "C" int pdgn_nndistance(int b, int n, const float *xyz, int m, const float *xyz2, float *result,
                               int32_t *result_i, float *result2, int32_t *result2_i,
                               pdgn_stream_t stream) {
    if (!sl_dims_ok(b, n, m)) return PDGN_ERR_INVALID;
    if (b == 0 || (n == 0 && m == 0)) return 0;
    dim3 grid(cdiv(n > m ? n : m, SL_THREADS), b, 2);
    hipLaunchKernelGGL(nndist_kernel, grid, dim3(SL_THREADS), 0, (hipStream_t)stream, n, m, xyz, xyz2,
                       result, result_i, result2, result2_i);
    return pdgn_launch_status();
}

extern "C" int pdgn_nndistance_grad(int b, int n, const float *xyz1, int m, const float *xyz2,
                                    const float *grad_dist1, const int32_t *idx1,
                                    const float *grad_dist2, const int32_t *idx2, float *grad_xyz1,
                                    float *grad_xyz2, pdgn_stream_t stream) {
    if (!sl_dims_ok(b, n, m)) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (n && (e = hipMemsetAsync(grad_xyz1, 0, (size_t)b * n * 3 * sizeof(float), s)) != hipSuccess) return (int)e;
    if (m && (e = hipMemsetAsync(grad_xyz2, 0, (size_t)b * m * 3 * sizeof(float), s)) != hipSuccess) return (int)e;
    if (n == 0 || m == 0) return 0;
    dim3 grid(cdiv(n > m ? n : m, SL_THREADS), b, 2);
    hipLaunchKernelGGL(nndist_grad_kernel, grid, dim3(SL_THREADS), 0, s, n, m, xyz1, xyz2, grad_dist1, idx1,
                       grad_dist2, idx2, grad_xyz1, grad_xyz2);
    return pdgn_launch_status();
}

static bool am_dims_ok(int b, int n, int m) {
    // one workgroup per pair; n, m >= 1 (the reference divides n/m or m/n)
    return b >= 0 && n >= 1 && m >= 1 && (long long)n * m <= 0x7fffffffLL;
}

extern "C" int pdgn_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                                float *temp, pdgn_stream_t stream) {
    if (!am_dims_ok(b, n, m)) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(approxmatch_kernel, dim3(b), dim3(AM_THREADS), 0, (hipStream_t)stream, n, m, xyz1, xyz2, match, temp);
    return pdgn_launch_status();
}

// floats of `temp` pdgn_emd_cost / pdgn_emd_cost_indexed need per call: per pair the four remain / ratio vectors of the
// reference's `temp` (2 (n + m), rounded up to a multiple of 4) and the two clouds as x-sorted float4 rows (4 (n + m))
extern "C" long long pdgn_emd_cost_temp_floats(long long pairs, int n, int m) {
    if (pairs < 0 || n < 1 || m < 1) return PDGN_ERR_INVALID;
    return pairs * ((((long long)(n + m) * 2 + 3) & ~3LL) + (long long)(n + m) * 4);
}

static void emd_launch(int pairs, int n, int m, const float *xyz1, const float *xyz2, float *temp, float *out, const int32_t *ia,
                       const int32_t *ib, hipStream_t s) {
    if (n <= 2048 && m <= 2048)                        // both clouds fit the LDS sort buffer
        hipLaunchKernelGGL(emd_cost_kernel<true>, dim3(pairs), dim3(AM_THREADS), 0, s, n, m, xyz1, xyz2, temp, out, ia, ib);
    else
        hipLaunchKernelGGL(emd_cost_kernel<false>, dim3(pairs), dim3(AM_THREADS), 0, s, n, m, xyz1, xyz2, temp, out, ia, ib);
}

extern "C" int pdgn_emd_cost(int b, int n, int m, const float *xyz1, const float *xyz2, float *temp,
                             float *out, pdgn_stream_t stream) {
    if (!am_dims_ok(b, n, m)) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    emd_launch(b, n, m, xyz1, xyz2, temp, out, nullptr, nullptr, (hipStream_t)stream);
    return pdgn_launch_status();
}

// Fused EMD cost of `npairs` (ia[p], ib[p]) cloud pairs drawn from xyz1 (., n, 3) and xyz2 (., m, 3).
extern "C" int pdgn_emd_cost_indexed(int npairs, int n, int m, const float *xyz1, const int32_t *ia,
                                     const float *xyz2, const int32_t *ib, float *temp, float *out,
                                     pdgn_stream_t stream) {
    if (!am_dims_ok(npairs, n, m)) return PDGN_ERR_INVALID;
    if (npairs == 0) return 0;
    emd_launch(npairs, n, m, xyz1, xyz2, temp, out, ia, ib, (hipStream_t)stream);
    return pdgn_launch_status();
}

extern "C" int pdgn_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2,
                              const float *match, float *out, pdgn_stream_t stream) {
    if (!am_dims_ok(b, n, m)) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipLaunchKernelGGL(matchcost_kernel, dim3(b), dim3(AM_THREADS), 0, (hipStream_t)stream, n, m, xyz1,
                       xyz2, match, out);
    return pdgn_launch_status();
}

extern "C" int pdgn_matchcost_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                                   const float *match, float *grad1, float *grad2,
                                   pdgn_stream_t stream) {
    if (!am_dims_ok(b, n, m) || b > 65535) return PDGN_ERR_INVALID;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(matchcost_grad1_kernel, dim3(cdiv(n, SL_THREADS), b), dim3(SL_THREADS), 0, s, n, m,
                       xyz1, xyz2, match, grad1);
    hipLaunchKernelGGL(matchcost_grad2_kernel, dim3(cdiv(m, SL_THREADS / PDGN_WAVE), b), dim3(SL_THREADS), 0,
                       s, n, m, xyz1, xyz2, match, grad2);
    return pdgn_launch_status();
}

extern "C" int pdgn_abi_version(void) { return 26; }
