// Cost, in matrix-pipe cycles, of the non-MFMA instructions a GEMM wave issues between its MFMAs (gfx950, one wave per
// SIMD, v_mfma_f32_16x16x4_f32 = 32 cycles): per group of 16 MFMAs, N instructions of one kind.
//   kind 0: nothing   1: ds_read_b128   2: buffer_load_dwordx4 (to VGPR)   3: buffer_load_dwordx4 ... lds (LDS-DMA)
//   4: s_waitcnt lgkmcnt(0) after the ds_reads   5: v_mov (VALU)   6: buffer_store_dwordx4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int N>
__global__ __launch_bounds__(256) void k(int iters, const float *in, float *out, unsigned long long *cyc) {
    __shared__ float sm[256 * 36];
    for (int i = threadIdx.x; i < 256 * 36; i += 256) sm[i] = in[i & 4095];
    __syncthreads();
    f32x4 acc[16];
    for (int a = 0; a < 16; ++a) acc[a] = (f32x4){0, 0, 0, 0};
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    const unsigned long long base = (unsigned long long)in;
    i32x4 srd;
    srd.x = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    srd.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu));
    srd.z = 1 << 20;
    srd.w = 0x00020000;
    const unsigned voff = threadIdx.x * 16u;
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)sm + (threadIdx.x >> 6) * 1024u * 8);
    const unsigned zero = __builtin_amdgcn_readfirstlane(0);
    f32x4 t[N > 0 ? N : 1];
    for (int i = 0; i < (N > 0 ? N : 1); ++i) t[i] = (f32x4){0, 0, 0, 0};
    float s = 0.f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
            if (a < N) {
                if (KIND == 1 || KIND == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[a]) : "v"(voff), "n"(a * 4096 % 32768));
                if (KIND == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(t[a]) : "v"(voff), "s"(srd));
                if (KIND == 3) asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds + a * 1024), "v"(voff), "s"(srd), "s"(zero) : "memory", "m0");
                if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(t[a].x) : "v"(voff));
                if (KIND == 6) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(acc[15 - a]), "v"(voff + (1u << 19)), "s"(srd) : "memory");
            }
        }
        if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == 2 || KIND == 3 || KIND == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = 0; i < N; ++i) s += t[i].x;
    }
    const unsigned long long t1 = clock64();
    for (int a = 0; a < 16; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float *in, *out;
    unsigned long long *cyc, h;
    hipMalloc(&in, 4 << 20);
    hipMemset(in, 0, 4 << 20);
    hipMalloc(&out, 1 << 22);
    hipMalloc(&cyc, 8);
    const char *names[7] = {"nothing", "ds_read_b128 (+1 wait)", "buffer_load_dwordx4 (+1 wait)", "LDS-DMA 1 KB (+1 wait)", "ds_read_b128 (+wait)", "v_mov_b32",
                            "buffer_store_dwordx4 (+1 wait)"};
    const int iters = 4000;
#define RUN(K, N)                                                                                   \
    hipLaunchKernelGGL((k<K, N>), dim3(256), dim3(256), 0, 0, iters, in, out, cyc);                 \
    hipDeviceSynchronize();                                                                         \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                   \
    printf("%-32s x%2d per 16 MFMA: %7.1f cycles per group (512 = matrix pipe alone)\n", names[K], N, (double)h / iters); \
    fflush(stdout);
    RUN(0, 0) RUN(1, 4) RUN(1, 8) RUN(1, 16) RUN(2, 2) RUN(2, 4) RUN(2, 8) RUN(3, 1) RUN(3, 2) RUN(3, 4) RUN(5, 4) RUN(5, 16) RUN(6, 4) RUN(6, 16)
    return 0;
}
