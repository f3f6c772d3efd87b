// Issue opportunities of the OTHER waves of a SIMD next to a back-to-back fp32 MFMA stream (gfx950): chains of v_fma with
// ILP independent chains interleaved, all finishing while the MFMA stream still runs.  MFMA shape 32x32x2 (64 cycles) or
// 16x16x4 (32 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ILP, int SHAPE, int YIELD = 0>
__global__ __launch_bounds__(512) void k(int iters, int citers, float *out, unsigned long long *cyc) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    if (wave < 4) {
        const unsigned long long t0 = clock64();
        if (SHAPE == 32) {
            f32x16 acc[4];
            for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
                    if (YIELD == 1) asm volatile("s_nop 0");
                    if (YIELD == 2) __builtin_amdgcn_s_sleep(1);
                    if (YIELD == 3) asm volatile("s_setprio 0");
                    if (YIELD == 4) asm volatile("s_nop 15");
                    if (YIELD == 5 && a == 3) __builtin_amdgcn_s_sleep(1);
                }
            for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        } else {
            f32x4 acc[8];
            for (int a = 0; a < 8; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
            for (int a = 0; a < 8; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
        }
        const unsigned long long t1 = clock64();
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    } else {
        float v[ILP];
        for (int c = 0; c < ILP; ++c) v[c] = x + c;
        const unsigned long long t0 = clock64();
        for (int it = 0; it < citers; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int c = 0; c < ILP; ++c) v[c] = __builtin_fmaf(v[c], y, x);
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
#define RUN(I, S)                                                                                      \
    hipLaunchKernelGGL((k<I, S>), dim3(256), dim3(512), 0, 0, 40000, 500, out, cyc);                   \
    hipDeviceSynchronize();                                                                            \
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);                                                      \
    printf("MFMA %dx%d: %5.1f cycles per MFMA | other wave, %d independent chains: %6.1f cycles per round of %d v_fma\n", S, S, \
           (double)h[0] / (40000.0 * (S == 32 ? 4 : 8)), I, (double)h[1] / 8000.0, I);                 \
    fflush(stdout);
    RUN(1, 32) RUN(8, 32) RUN(1, 16)
#define RUNY(I, Y)                                                                                     \
    hipLaunchKernelGGL((k<I, 32, Y>), dim3(256), dim3(512), 0, 0, 40000, 500, out, cyc);               \
    hipDeviceSynchronize();                                                                            \
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);                                                      \
    printf("yield variant %d: %5.1f cycles per MFMA | other wave, %d chains: %6.1f cycles per round\n", Y, (double)h[0] / 160000.0, I, \
           (double)h[1] / 8000.0);                                                                     \
    fflush(stdout);
    RUNY(1, 1) RUNY(1, 2) RUNY(1, 3) RUNY(1, 4) RUNY(1, 5) RUNY(8, 2) RUNY(8, 5)
    return 0;
}
