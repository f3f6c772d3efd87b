// Achievable fp32 MFMA rate of this chip: v_mfma_f32_32x32x2_f32 chains with no memory traffic.
// usage: mfma_peak [waves_per_simd=2] [iters=20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float *out) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 20000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * wps;
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(256), 0, 0, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 8 * 4096.0;
        printf("CUs %d clock %d MHz  waves/SIMD %d  %.3f ms  %.1f TFLOP/s\n", p.multiProcessorCount, p.clockRate / 1000, wps, ms,
               flops / ms / 1e9);
    }
    return 0;
}
