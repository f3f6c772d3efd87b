// small_mlp.hip -- Linear (+ BatchNorm1d) (+ LeakyReLU / ReLU) on a handful of rows in ONE launch.
// The generator's global branches and its first layer (models/PDGNet_v2.py:704-707, 825-828) apply
// nn.Sequential(Linear, BatchNorm1d, LeakyReLU) to one vector per SAMPLE: x is (B, K) with B = 35.  As library calls
// that is GEMM + statistics + transform + counter + activation forward and six more launches backward, each a few
// microseconds of work behind tens of microseconds of launch latency, on the critical path of every block.
// Here a wave owns one output channel and a lane one row: the BatchNorm statistics of a channel are a wave
// reduction, x sits in LDS, W[n, :] is a wave-uniform stream.  Rows <= 64, K <= SM_MAXK.
#include "common.h"

#define SM_THREADS 256
#define SM_MAXK 1024

__device__ __forceinline__ float sm_act(float z, int act) { return act == 2 ? (z > 0.f ? z : 0.01f * z) : (act == 1 ? fmaxf(z, 0.f) : z); }
__device__ __forceinline__ float sm_actg(float z, int act) { return act == 2 ? (z > 0.f ? 1.f : 0.01f) : (act == 1 ? (z > 0.f ? 1.f : 0.f) : 1.f); }
__device__ __forceinline__ float sm_wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// y = act(BN(x W^T + b)).  bn_mode: 0 none, 1 batch statistics (+ running update), 2 running statistics.
// pre (R,N): the linear output, stat (2N): [mean | invstd] -- saved for the adjoint (may be NULL when not needed).
__global__ __launch_bounds__(SM_THREADS) void small_mlp_fwd_kernel(
    int R, int K, int N, int act, int bn_mode, float eps, float momentum, const float *__restrict__ x,
    const float *__restrict__ W, const float *__restrict__ bias, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ running_mean, float *__restrict__ running_var,
    float *__restrict__ y, float *__restrict__ pre, float *__restrict__ stat) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [R][ld]
    // rows padded to a multiple of 4 floats + 4: 16-B aligned rows (ds_read_b128) whose lane stride is 4 banks off a multiple
    // of 64 (conflict-free); K % 4 != 0 (never on the model's layers) keeps the scalar form
    const bool vec = (K & 3) == 0;
    const int ld = vec ? K + 4 : K + 1;
    if (vec) {
        const int K4 = K >> 2;
        for (int i = threadIdx.x; i < R * K4; i += SM_THREADS) {
            const int r = i / K4, c = i - r * K4;
            *reinterpret_cast<float4 *>(xs + r * ld + 4 * c) = *reinterpret_cast<const float4 *>(x + (size_t)r * K + 4 * c);
        }
    } else {
        for (int i = threadIdx.x; i < R * K; i += SM_THREADS) xs[(i / K) * ld + i % K] = x[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform: W[n, :] becomes a scalar-load stream
    const int n = blockIdx.x * (SM_THREADS / 64) + wave;
    if (n >= N) return;
    const bool live = lane < R;
    const float *w = W + (size_t)n * K;
    const float *xr = xs + (live ? lane : 0) * ld;
    float acc = bias ? bias[n] : 0.f;
    if (vec) {
#pragma unroll 8
        for (int k = 0; k < K; k += 4) {                         // the same fma chain, k ascending: bit-identical to the scalar form
                                                                 // (unrolled x8: 32 k of LDS / scalar loads in flight per wait)
            const float4 xv = *reinterpret_cast<const float4 *>(xr + k);
            const float4 wv = *reinterpret_cast<const float4 *>(w + k);
            acc = __fmaf_rn(xv.x, wv.x, acc);
            acc = __fmaf_rn(xv.y, wv.y, acc);
            acc = __fmaf_rn(xv.z, wv.z, acc);
            acc = __fmaf_rn(xv.w, wv.w, acc);
        }
    } else {
        for (int k = 0; k < K; ++k) acc = __fmaf_rn(xr[k], w[k], acc);
    }
    if (pre && live) pre[(size_t)lane * N + n] = acc;
    float z = acc;
    if (bn_mode) {
        float mean, invstd;
        if (bn_mode == 1) {
            mean = sm_wsum(live ? acc : 0.f) / (float)R;
            const float d = live ? acc - mean : 0.f;
            const float var = sm_wsum(d * d) / (float)R;
            invstd = rsqrtf(var + eps);
            if (lane == 0 && running_mean) {
                running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * mean;
                running_var[n] = (1.f - momentum) * running_var[n] + momentum * (R > 1 ? var * (float)R / (float)(R - 1) : var);
            }
        } else {
            mean = running_mean[n];
            invstd = rsqrtf(running_var[n] + eps);
        }
        if (stat && lane == 0) { stat[n] = mean; stat[N + n] = invstd; }
        z = (acc - mean) * invstd * (gamma ? gamma[n] : 1.f) + (beta ? beta[n] : 0.f);
    }
    if (live) y[(size_t)lane * N + n] = sm_act(z, act);
}

// Adjoint wrt the linear output, the BatchNorm parameters, the bias and W:
//   dpre (R,N), dgamma (N), dbeta (N), dbias (N), dW (N,K)   (any of the last four may be NULL).
__global__ __launch_bounds__(SM_THREADS) void small_mlp_bwd_kernel(
    int R, int K, int N, int act, int bn_mode, const float *__restrict__ x, const float *__restrict__ dy,
    const float *__restrict__ pre, const float *__restrict__ stat, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ dpre, float *__restrict__ dgamma, float *__restrict__ dbeta,
    float *__restrict__ dbias, float *__restrict__ dW) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // xs [R][ld] | dp [4][64]
    const bool vec = (K & 3) == 0;                              // 16-B rows as in the forward kernel
    const int ld = vec ? K + 4 : K + 1;
    float *xs = sm, *dps = sm + (size_t)R * ld;
    if (vec) {
        const int K4 = K >> 2;
        for (int i = threadIdx.x; i < R * K4; i += SM_THREADS) {
            const int r = i / K4, c = i - r * K4;
            *reinterpret_cast<float4 *>(xs + r * ld + 4 * c) = *reinterpret_cast<const float4 *>(x + (size_t)r * K + 4 * c);
        }
    } else {
        for (int i = threadIdx.x; i < R * K; i += SM_THREADS) xs[(i / K) * ld + i % K] = x[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x * (SM_THREADS / 64) + wave;
    if (n >= N) return;
    const bool live = lane < R;
    const float p = live ? pre[(size_t)lane * N + n] : 0.f;
    const float g = live ? dy[(size_t)lane * N + n] : 0.f;
    float dp;
    if (bn_mode) {
        const float mean = stat[n], invstd = stat[N + n], ga = gamma ? gamma[n] : 1.f, be = beta ? beta[n] : 0.f;
        const float xhat = (p - mean) * invstd;
        const float dz = live ? g * sm_actg(xhat * ga + be, act) : 0.f;
        const float s1 = sm_wsum(dz), s2 = sm_wsum(dz * xhat);
        if (lane == 0) {
            if (dbeta) dbeta[n] = s1;
            if (dgamma) dgamma[n] = s2;
        }
        dp = bn_mode == 1 ? ga * invstd * (dz - s1 / (float)R - xhat * (s2 / (float)R)) : ga * invstd * dz;
    } else {
        dp = g * sm_actg(p, act);
    }
    if (!live) dp = 0.f;
    if (live) dpre[(size_t)lane * N + n] = dp;
    const float sb = sm_wsum(dp);
    if (dbias && lane == 0) dbias[n] = sb;
    if (dW) {
        dps[wave * 64 + lane] = dp;
        __builtin_amdgcn_wave_barrier();
        if (vec) {                                               // four columns per lane: one 16-B LDS read per row (same sums, same order)
            for (int k = 4 * lane; k < K; k += 256) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 5
                for (int r = 0; r < R; ++r) {
                    const float d = dps[wave * 64 + r];
                    const float4 xv = *reinterpret_cast<const float4 *>(xs + r * ld + k);
                    s.x = __fmaf_rn(d, xv.x, s.x);
                    s.y = __fmaf_rn(d, xv.y, s.y);
                    s.z = __fmaf_rn(d, xv.z, s.z);
                    s.w = __fmaf_rn(d, xv.w, s.w);
                }
                *reinterpret_cast<float4 *>(dW + (size_t)n * K + k) = s;
            }
        } else {
            for (int k = lane; k < K; k += 64) {
                float s = 0.f;
                for (int r = 0; r < R; ++r) s = __fmaf_rn(dps[wave * 64 + r], xs[r * ld + k], s);
                dW[(size_t)n * K + k] = s;
            }
        }
    }
}

static bool sm_ok(int r, int k, int n, int act, int bn_mode) {
    return r >= 1 && r <= 64 && k >= 1 && k <= SM_MAXK && n >= 1 && act >= 0 && act <= 2 && bn_mode >= 0 && bn_mode <= 2 &&
           (size_t)r * (k + 4) * 4 + 1024 <= 160 * 1024;
}

extern "C" int pdgn_small_mlp_forward(int r, int k, int n, int act, int bn_mode, float eps, float momentum, const float *x,
                                      const float *W, const float *bias, const float *gamma, const float *beta,
                                      float *running_mean, float *running_var, float *y, float *pre, float *stat,
                                      pdgn_stream_t stream) {
    if (!sm_ok(r, k, n, act, bn_mode) || (bn_mode == 2 && (!running_mean || !running_var))) return PDGN_ERR_INVALID;
    const size_t lds = (size_t)r * (k + 4) * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)small_mlp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(small_mlp_fwd_kernel, dim3(cdiv(n, SM_THREADS / 64)), dim3(SM_THREADS), lds, (hipStream_t)stream, r, k, n, act,
                       bn_mode, eps, momentum, x, W, bias, gamma, beta, running_mean, running_var, y, pre, stat);
    return pdgn_launch_status();
}

extern "C" int pdgn_small_mlp_backward(int r, int k, int n, int act, int bn_mode, const float *x, const float *dy,
                                       const float *pre, const float *stat, const float *gamma, const float *beta,
                                       float *dpre, float *dgamma, float *dbeta, float *dbias, float *dW,
                                       pdgn_stream_t stream) {
    if (!sm_ok(r, k, n, act, bn_mode) || (bn_mode && !stat)) return PDGN_ERR_INVALID;
    const size_t lds = ((size_t)r * (k + 4) + 4 * 64) * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)small_mlp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(small_mlp_bwd_kernel, dim3(cdiv(n, SM_THREADS / 64)), dim3(SM_THREADS), lds, (hipStream_t)stream, r, k, n, act,
                       bn_mode, x, dy, pre, stat, gamma, beta, dpre, dgamma, dbeta, dbias, dW);
    return pdgn_launch_status();
}
