// adam.hip -- the Adam update of a whole parameter list in ONE launch with thousands of workgroups.
//
// The reference steps five torch.optim.Adam optimisers per iteration (models/PDGNet_v2.py:121-125, 186-226, 256: lr 1e-4,
// betas (0.5, 0.999), no weight decay, no amsgrad).  torch's fused multi-tensor kernel walks a list in chunks of 64 K elements --
// 194 workgroups for the generator's 12.7 M parameters, in five launches: 230 us at the end of every iteration, on the chain the next
// iteration waits for (1.5 TB/s of the 355 MB it moves).  Here a workgroup takes ADAM_CHUNK elements of one tensor of the list
// (found by bisection in a table of first chunks); the arithmetic is torch's (ATen/native/cuda/fused_adam_utils.cuh, the
// non-amsgrad, non-maximize, weight_decay = 0 case):
//     m <- beta1 m + (1 - beta1) g;  v <- beta2 v + (1 - beta2) g g;        (in fp64, as torch's double betas make them)
//     p <- p - (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// with the bias corrections in fp64 from the step count t that torch keeps as a device tensor (read here, incremented by the caller).
#include "common.h"

#define ADAM_THREADS 256
#define ADAM_CHUNK 4096            // elements per workgroup: four float4 per thread
#define ADAM_MAXT 72               // tensors per launch: their pointers travel in the kernel arguments (3.5 KB of the 4 KB there are)

struct AdamArgs {                  // by value: a recorded iteration (csrc/replay.hip) re-issues the launch with the same pointers
    float *p[ADAM_MAXT];
    const float *g[ADAM_MAXT];
    float *m[ADAM_MAXT];
    float *v[ADAM_MAXT];
    long long n[ADAM_MAXT];        // elements
    int chunk0[ADAM_MAXT];         // index of the tensor's first chunk among this launch's chunks
    int ntensors;
};

__global__ __launch_bounds__(ADAM_THREADS) void adam_multi_kernel(const AdamArgs a, double lr, double beta1d, double beta2d, double eps,
                                                                  const float *__restrict__ step) {
    __shared__ float sc[2];
    // the chunk's tensor: the last one whose first chunk is <= this chunk
    int lo = 0, hi = a.ntensors - 1;
    const int c = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.chunk0[mid] <= c) lo = mid; else hi = mid - 1;
    }
    float *const P = a.p[lo], *const M = a.m[lo], *const V = a.v[lo];
    const float *const G = a.g[lo];
    const long long n = a.n[lo];
    if (threadIdx.x == 0) {
        // torch: the two bias corrections in fp64, handed to the arithmetic as floats; step_size = lr (double) / that float
        const double t = (double)step[0];
        const float bc1 = (float)(1.0 - pow(beta1d, t));
        sc[0] = (float)(lr / (double)bc1);
        sc[1] = (float)sqrt(1.0 - pow(beta2d, t));
    }
    __syncthreads();
    const float step_size = sc[0], bc2s = sc[1];
    const double epsd = (double)eps, w1 = 1.0 - beta1d, w2 = 1.0 - beta2d;
    const long long i0 = (long long)(c - a.chunk0[lo]) * ADAM_CHUNK;
    const bool vec = ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0);
    // torch's expressions with torch's types (lr, betas, eps are doubles there: the moment updates are evaluated in fp64 and rounded
    // once, 1 - beta is 1 - the DOUBLE beta): the results are torch's bits, not merely close to them
    auto one = [&](float &p, float g, float &m, float &v) {
        m = (float)(beta1d * (double)m + w1 * (double)g);
        v = (float)(beta2d * (double)v + (w2 * (double)g) * (double)g);
        const float denom = (float)((double)(sqrtf(v) / bc2s) + epsd);
        p -= step_size * m / denom;
    };
    if (vec) {
        // all loads of the chunk first (the stores below may alias them as far as the compiler knows: interleaved, every float4
        // group waited for the one before), then the arithmetic, then the stores
        constexpr int NU = ADAM_CHUNK / (4 * ADAM_THREADS);
        float4 p4[NU], g4[NU], m4[NU], v4[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const long long i = i0 + 4LL * (threadIdx.x + u * ADAM_THREADS);
            if (i + 3 < n) {
                p4[u] = *reinterpret_cast<const float4 *>(P + i);
                g4[u] = *reinterpret_cast<const float4 *>(G + i);
                m4[u] = *reinterpret_cast<const float4 *>(M + i);
                v4[u] = *reinterpret_cast<const float4 *>(V + i);
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const long long i = i0 + 4LL * (threadIdx.x + u * ADAM_THREADS);
            if (i + 3 < n) {
                one(p4[u].x, g4[u].x, m4[u].x, v4[u].x); one(p4[u].y, g4[u].y, m4[u].y, v4[u].y);
                one(p4[u].z, g4[u].z, m4[u].z, v4[u].z); one(p4[u].w, g4[u].w, m4[u].w, v4[u].w);
                *reinterpret_cast<float4 *>(P + i) = p4[u];
                *reinterpret_cast<float4 *>(M + i) = m4[u];
                *reinterpret_cast<float4 *>(V + i) = v4[u];
            } else {
                for (long long j = i; j < n && j < i + 4; ++j) one(P[j], G[j], M[j], V[j]);
            }
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < n && i < i0 + ADAM_CHUNK; i += ADAM_THREADS) one(P[i], G[i], M[i], V[i]);
    }
}

// One Adam step of `ntensors` fp32 tensors (p, g, m, v: HOST arrays of device pointers; n: their element counts), in
// ceil(ntensors / 72) launches of one workgroup per 4096 elements.  step (device): the step count t >= 1 of THIS update as one
// float (torch's `state["step"]` after its increment).  Replaces torch._fused_adam_ / optimizer.step() of the reference's five Adam
// optimisers (models/PDGNet_v2.py:121-125) for lists without weight decay, amsgrad or maximize.
extern "C" int pdgn_adam_multi(int ntensors, void *const *p, const void *const *g, void *const *m, void *const *v, const long long *n,
                               double lr, double beta1, double beta2, double eps, const float *step, pdgn_stream_t stream) {
    if (ntensors < 1 || !p || !g || !m || !v || !n || !step || !(lr >= 0.) || !(beta1 >= 0. && beta1 < 1.) ||
        !(beta2 >= 0. && beta2 < 1.) || !(eps >= 0.))
        return PDGN_ERR_INVALID;
    for (int i = 0; i < ntensors; ++i)
        if (!p[i] || !g[i] || !m[i] || !v[i] || n[i] < 1 || (((uintptr_t)p[i] | (uintptr_t)g[i] | (uintptr_t)m[i] | (uintptr_t)v[i]) & 3))
            return PDGN_ERR_INVALID;
    for (int t0 = 0; t0 < ntensors; t0 += ADAM_MAXT) {
        AdamArgs a;
        a.ntensors = ntensors - t0 < ADAM_MAXT ? ntensors - t0 : ADAM_MAXT;
        long long chunks = 0;
        for (int i = 0; i < a.ntensors; ++i) {
            a.p[i] = (float *)p[t0 + i]; a.g[i] = (const float *)g[t0 + i]; a.m[i] = (float *)m[t0 + i]; a.v[i] = (float *)v[t0 + i];
            a.n[i] = n[t0 + i];
            a.chunk0[i] = (int)chunks;
            chunks += (n[t0 + i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
            if (chunks > 0x3fffffffLL) return PDGN_ERR_INVALID;
        }
        for (int i = a.ntensors; i < ADAM_MAXT; ++i) { a.p[i] = a.m[i] = a.v[i] = nullptr; a.g[i] = nullptr; a.n[i] = 0; a.chunk0[i] = 0x7fffffff; }
        hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)chunks), dim3(ADAM_THREADS), 0, (hipStream_t)stream, a, lr, beta1, beta2, eps, step);
    }
    return pdgn_launch_status();
}

// ---- the same walk for a plain copy: dst[i] <- src[i] for a list of fp32 tensors (the pack of a network's fresh gradients into the
// flat all-reduce buffer, trainer.FlatGrads: torch._foreach_copy_ takes 82 us for the generator's 160 tensors / 50.8 MB)
#define COPY_MAXT 128               // tensors per launch (28 bytes of arguments each)
struct CopyArgs {
    float *d[COPY_MAXT];
    const float *s[COPY_MAXT];
    long long n[COPY_MAXT];
    int chunk0[COPY_MAXT];
    int ntensors;
};

__global__ __launch_bounds__(ADAM_THREADS) void copy_multi_kernel(const CopyArgs a) {
    int lo = 0, hi = a.ntensors - 1;
    const int c = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.chunk0[mid] <= c) lo = mid; else hi = mid - 1;
    }
    float *const D = a.d[lo];
    const float *const S = a.s[lo];
    const long long n = a.n[lo], i0 = (long long)(c - a.chunk0[lo]) * ADAM_CHUNK;
    if (((((uintptr_t)D | (uintptr_t)S) & 15) == 0)) {
        constexpr int NU = ADAM_CHUNK / (4 * ADAM_THREADS);
        float4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const long long i = i0 + 4LL * (threadIdx.x + u * ADAM_THREADS);
            if (i + 3 < n) v[u] = *reinterpret_cast<const float4 *>(S + i);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const long long i = i0 + 4LL * (threadIdx.x + u * ADAM_THREADS);
            if (i + 3 < n) *reinterpret_cast<float4 *>(D + i) = v[u];
            else for (long long j = i; j < n && j < i + 4; ++j) D[j] = S[j];
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < n && i < i0 + ADAM_CHUNK; i += ADAM_THREADS) D[i] = S[i];
    }
}

// dst[i] (n[i] floats) <- src[i] for ntensors fp32 tensors (HOST arrays of device pointers, 4-byte aligned; 16-byte aligned pairs
// take the vector path), in ceil(ntensors / 128) launches.  Replaces torch._foreach_copy_ where a network's gradients are packed
// into one buffer for the all-reduce (the DataParallel gradient reduction of models/PDGNet_v2.py:101-105).
extern "C" int pdgn_copy_multi(int ntensors, void *const *dst, const void *const *src, const long long *n, pdgn_stream_t stream) {
    if (ntensors < 1 || !dst || !src || !n) return PDGN_ERR_INVALID;
    for (int i = 0; i < ntensors; ++i)
        if (!dst[i] || !src[i] || n[i] < 1 || (((uintptr_t)dst[i] | (uintptr_t)src[i]) & 3)) return PDGN_ERR_INVALID;
    for (int t0 = 0; t0 < ntensors; t0 += COPY_MAXT) {
        CopyArgs a;
        a.ntensors = ntensors - t0 < COPY_MAXT ? ntensors - t0 : COPY_MAXT;
        long long chunks = 0;
        for (int i = 0; i < a.ntensors; ++i) {
            a.d[i] = (float *)dst[t0 + i]; a.s[i] = (const float *)src[t0 + i]; a.n[i] = n[t0 + i];
            a.chunk0[i] = (int)chunks;
            chunks += (n[t0 + i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
            if (chunks > 0x3fffffffLL) return PDGN_ERR_INVALID;
        }
        for (int i = a.ntensors; i < COPY_MAXT; ++i) { a.d[i] = nullptr; a.s[i] = nullptr; a.n[i] = 0; a.chunk0[i] = 0x7fffffff; }
        hipLaunchKernelGGL(copy_multi_kernel, dim3((unsigned)chunks), dim3(ADAM_THREADS), 0, (hipStream_t)stream, a);
    }
    return pdgn_launch_status();
}
