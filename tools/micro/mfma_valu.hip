// How fp32 MFMA and ordinary vector instructions share a SIMD on gfx950 (measurement behind the feature-kNN design).
//  mode 0: every wave issues v_mfma_f32_32x32x2_f32 back to back with NV independent v_fma_f32 after each MFMA.
//  mode 1: 8 waves per workgroup (2 per SIMD): waves 0-3 MFMA only, waves 4-7 a DEPENDENT chain of v_fma_f32 only;
//          both report their own cycle counts (prio: s_setprio of the VALU waves).
// usage: mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void same_wave(int iters, float *out, unsigned long long *cyc) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = __builtin_fmaf(v[i], y, x);       // NV independent chains
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int PRIO, int SHAPE, int ILP = 1, int VW = 4>
__global__ __launch_bounds__(256 + 64 * VW) void two_waves(int iters, int viters, float *out, unsigned long long *cyc) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    if (wave < 4) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        const unsigned long long t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
            if (SHAPE == 1) asm volatile("s_nop 0");
            if (SHAPE == 2) __builtin_amdgcn_s_sleep(1);
        }
        const unsigned long long t1 = clock64();
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        float v[ILP];
        for (int c = 0; c < ILP; ++c) v[c] = x + c;
        const unsigned long long t0 = clock64();
        for (int it = 0; it < viters; ++it) {
#pragma unroll
            for (int i = 0; i < 16 / ILP; ++i)
#pragma unroll
                for (int c = 0; c < ILP; ++c) v[c] = __builtin_fmaf(v[c], y, x);   // ILP dependent chains, interleaved
        }
        const unsigned long long t1 = clock64();
        for (int c = 0; c < ILP; ++c) s += v[c];
        if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *out;
    unsigned long long *cyc, h[2];
    hipMalloc(&out, 1 << 24);
    hipMalloc(&cyc, 16);
    const int iters = 20000;
#define SAME(NV)                                                                                          \
    hipLaunchKernelGGL(same_wave<NV>, dim3(256), dim3(256), 0, 0, iters, out, cyc);                       \
    hipDeviceSynchronize();                                                                               \
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);                                                         \
    printf("same wave: %2d v_fma per MFMA: %6.1f cycles per MFMA\n", NV, (double)h[0] / (iters * 4.0));
    SAME(0) SAME(0) SAME(1) SAME(2) SAME(4) SAME(8) SAME(12) SAME(16)
#define TWO(P, S)                                                                                         \
    hipLaunchKernelGGL((two_waves<P, S, 1, 4>), dim3(256), dim3(512), 0, 0, iters, iters, out, cyc);            \
    hipDeviceSynchronize();                                                                               \
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);                                                         \
    printf("two waves per SIMD (valu prio %d, mfma loop variant %d): %6.1f cycles per MFMA, %6.1f cycles per dependent v_fma\n", P, S, \
           (double)h[0] / (iters * 4.0), (double)h[1] / (iters * 16.0));
    TWO(0, 0) TWO(0, 0) TWO(3, 0) TWO(0, 1) TWO(0, 2)
#define TWOI(I, W)                                                                                        \
    hipLaunchKernelGGL((two_waves<0, 0, I, W>), dim3(256), dim3(256 + 64 * W), 0, 0, iters, iters, out, cyc);  \
    hipDeviceSynchronize();                                                                               \
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);                                                         \
    printf("%d VALU waves next to 4 MFMA waves, %d chains per wave: %6.1f cycles per MFMA, %6.1f cycles per v_fma of a wave\n", W, I, \
           (double)h[0] / (iters * 4.0), (double)h[1] / (iters * 16.0));
    TWOI(2, 4) TWOI(4, 4) TWOI(8, 4) TWOI(1, 8) TWOI(4, 8) TWOI(8, 8) TWOI(4, 12)
    // VALU waves alone (no MFMA iterations)
    hipLaunchKernelGGL((two_waves<0, 0, 1, 4>), dim3(256), dim3(512), 0, 0, 0, iters, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("valu waves alone: %6.1f cycles per dependent v_fma\n", (double)h[1] / (iters * 16.0));
    return 0;
}
