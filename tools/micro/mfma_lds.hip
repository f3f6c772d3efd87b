// Ceiling of an fp32 MFMA inner loop that re-reads its operands from LDS (no global traffic, no barriers):
// wave tile 64x64, operands by ds_read_b128 (16 k per read), v_mfma_f32_16x16x4_f32 or v_mfma_f32_32x32x2_f32.
// usage: mfma_lds [wgs_per_cu=2] [iters=4000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 16: 16x16x4 (4x4 accumulators), 32: 32x32x2 (2x2 accumulators)
__global__ __launch_bounds__(256) void loop(int iters, const float *in, float *out) {
    __shared__ float sm[256 * 32];
    for (int i = threadIdx.x; i < 256 * 32; i += 256) sm[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
        const int li = lane & 15, lg = lane >> 4;
        const float *sa = sm + (wave >> 1) * 64 * 32, *sb = sm + (128 + (wave & 1) * 64) * 32;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float4 av[4], bv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) av[a] = *(const float4 *)(sa + (a * 16 + li) * 32 + 4 * ((4 * q + lg) ^ (li & 7)));
#pragma unroll
                for (int b = 0; b < 4; ++b) bv[b] = *(const float4 *)(sb + (b * 16 + li) * 32 + 4 * ((4 * q + lg) ^ (li & 7)));
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(((float *)&bv[b])[u], ((float *)&av[a])[u], acc[a][b], 0, 0, 0);
            }
        }
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
    } else {
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        const int li = lane & 31, lh = lane >> 5;
        const float *sa = sm + (wave >> 1) * 64 * 32, *sb = sm + (128 + (wave & 1) * 64) * 32;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 av[2], bv[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) av[a] = *(const float4 *)(sa + (a * 32 + li) * 32 + 4 * ((2 * q + lh) ^ (li & 7)));
#pragma unroll
                for (int b = 0; b < 2; ++b) bv[b] = *(const float4 *)(sb + (b * 32 + li) * 32 + 4 * ((2 * q + lh) ^ (li & 7)));
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(((float *)&bv[b])[u], ((float *)&av[a])[u], acc[a][b], 0, 0, 0);
            }
        }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 4000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * wpc;
    float *in, *out;
    hipMalloc(&in, 256 * 32 * 4);
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    float *h = (float *)malloc(256 * 32 * 4);
    for (int i = 0; i < 256 * 32; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h, 256 * 32 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        const int shape = rep & 1 ? 32 : 16;
        hipEventRecord(e0);
        if (shape == 16) hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(256), 0, 0, iters, in, out);
        else hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(256), 0, 0, iters, in, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 2.0 * 64 * 64 * 32;
        printf("shape %dx%d  WGs/CU %d  %.3f ms  %.1f TFLOP/s\n", shape, shape, wpc, ms, flops / ms / 1e9);
    }
    return 0;
}
