// What another wave of the SIMD gets done next to a back-to-back fp32 MFMA stream on gfx950: dependent chains of
// (0) v_fma, (1) s_add, (2) ds_read (pointer chase), (3) v_cmp -> s_bcnt1 -> v_add (ballot round trip), (4) ds_bpermute.
// 4 MFMA waves + 4 chain waves per workgroup, one workgroup per CU; reports cycles per chain step with and without MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void k(int iters, int citers, float *out, unsigned long long *cyc) {
    __shared__ int chase[1024];
    for (int i = threadIdx.x; i < 1024; i += 512) chase[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    if (wave < 4) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        const unsigned long long t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
        }
        const unsigned long long t1 = clock64();
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    } else {
        const unsigned long long t0 = clock64();
        if (KIND == 0) {
            float v = x;
            for (int it = 0; it < citers; ++it)
#pragma unroll
                for (int i = 0; i < 16; ++i) v = __builtin_fmaf(v, y, x);
            s = v;
        } else if (KIND == 1) {
            int a = __builtin_amdgcn_readfirstlane(citers);
            for (int it = 0; it < citers; ++it)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("s_add_i32 %0, %0, 3\n s_xor_b32 %0, %0, 0x55" : "+s"(a));
            s = a;
        } else if (KIND == 2) {
            int p = lane;
            for (int it = 0; it < citers; ++it)
#pragma unroll
                for (int i = 0; i < 16; ++i) p = chase[p];
            s = p;
        } else if (KIND == 3) {
            int v = lane;
            for (int it = 0; it < citers; ++it)
#pragma unroll
                for (int i = 0; i < 16; ++i) v += __popcll(__ballot(v & 1)) + 1;
            s = v;
        } else {
            int v = lane;
            for (int it = 0; it < citers; ++it)
#pragma unroll
                for (int i = 0; i < 16; ++i) v = __shfl_xor(v, 1, 64) + 1;
            s = v;
        }
        const unsigned long long t1 = clock64();
        if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    float *out;
    unsigned long long *cyc, h[2];
    hipMalloc(&out, 1 << 24);
    hipMalloc(&cyc, 16);
    const char *names[5] = {"v_fma", "s_add+s_xor", "ds_read chase", "v_cmp->s_bcnt->v_add", "ds_bpermute+add"};
#define RUN(K)                                                                                         \
    for (int with = 1; with >= 0 && (only < 0 || only == K); --with) {                                                            \
        hipLaunchKernelGGL(k<K>, dim3(256), dim3(512), 0, 0, with ? 4000 : 0, 1000, out, cyc);         \
        hipDeviceSynchronize();                                                                        \
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);                                                  \
        printf("%-22s %s MFMA stream: %6.1f cycles per step%s\n", names[K], with ? "next to" : "without", (double)h[1] / 16000.0, \
               with ? "" : "\n");                                                                      \
        fflush(stdout);                                                                                \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    return 0;
}
