// split_mfma_probe.hip -- can one wave split fp32 fragments into three bf16 parts (VALU) while its bf16 MFMAs run?
//
// fp32 products on the bf16 matrix cores: x = h + m + l with h, m, l bf16 (round to nearest each time, l exact to 2^-25 |x|),
// a b ~= ah bh + (ah bm + am bh) + (ah bl + am bm + al bh): six v_mfma_f32_16x16x32_bf16 per fp32 16x16x32 product,
// dropped terms <= 2^-23 |a b|.  The fp32 MFMA (v_mfma_f32_16x16x4_f32) needs 8 x 32 = 256 cycles for the same product,
// six bf16 MFMAs 6 x 16 = 96 -- if the splitting is free.  This probe runs the inner loop of a 4 x 4-fragment wave tile out
// of LDS only (no global traffic) in three forms and prints fp32-equivalent TFLOP/s of the whole chip:
//   mode 0: operands already bf16 in registers (the MFMA ceiling);
//   mode 1: the weight fragments read as bf16 parts from LDS, the activation fragments read as fp32 and split in the wave;
//   mode 2: both read as fp32 and split in the wave.
//   build: hipcc -O3 --offload-arch=gfx950 tools/split_mfma_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Parts {
    u32x4 h, m, l;
};

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// eight fp32 -> three bf16x8 (as 4 dwords each)
__device__ __forceinline__ Parts split8(const float4 x0, const float4 x1) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    Parts p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned h = cvt_pk(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const unsigned m = cvt_pk(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        p.h[i] = h;
        p.m[i] = m;
        p.l[i] = cvt_pk(sa, sb);
    }
    return p;
}

__device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int MODE, int TM, int TN>
__global__ __launch_bounds__(256, 2) void probe(float *out, int iters) {
    __shared__ float4 lds[4096];                                   // 64 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = make_float4(1.0f + i * 1e-4f, 0.5f - i * 1e-5f, 0.25f, -0.75f + i * 1e-6f);
    __syncthreads();
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Parts w[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) w[b] = split8(lds[lane + 64 * b], lds[lane + 64 * b + 1]);
    Parts x0 = split8(lds[lane], lds[lane + 7]);
    for (int it = 0; it < iters; ++it) {
        const int base = (it & 7) * 256;
        if (MODE == 1) {
#pragma unroll
            for (int b = 0; b < TN; ++b) {                         // bf16 parts straight from LDS: 3 x 16 B per fragment
                w[b].h = __builtin_bit_cast(u32x4, lds[base + lane + 64 * b]);
                w[b].m = __builtin_bit_cast(u32x4, lds[base + 1024 + lane + 64 * b]);
                w[b].l = __builtin_bit_cast(u32x4, lds[base + 2048 + lane + 64 * b]);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int b = 0; b < TN; ++b) w[b] = split8(lds[base + 2 * lane + 128 * b], lds[base + 2 * lane + 128 * b + 1]);
        }
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            Parts x = x0;
            if (MODE != 0) x = split8(lds[base + 2048 + 2 * lane + 128 * a], lds[base + 2048 + 2 * lane + 128 * a + 1]);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                f32x4 c = acc[a][b];
                c = mfma(w[b].l, x.h, c);                          // smallest terms first
                c = mfma(w[b].m, x.m, c);
                c = mfma(w[b].h, x.l, c);
                c = mfma(w[b].m, x.h, c);
                c = mfma(w[b].h, x.m, c);
                c = mfma(w[b].h, x.h, c);
                acc[a][b] = c;
            }
        }
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) s += acc[a][b];
    *reinterpret_cast<f32x4 *>(out + ((size_t)blockIdx.x * 256 + tid) * 4) = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the same loop on v_mfma_f32_32x32x16_bf16: TM x TN fragments of 32 x 32, two 16-deep k steps per 32-deep chunk -- a fragment
// element feeds 32 outputs instead of 16, so the wave splits (and reads from LDS) half as many elements per flop
template <int MODE, int TM, int TN>
__global__ __launch_bounds__(256, 2) void probe32(float *out, int iters) {
    __shared__ float4 lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = make_float4(1.0f + i * 1e-4f, 0.5f - i * 1e-5f, 0.25f, -0.75f + i * 1e-6f);
    __syncthreads();
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    Parts w[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) w[b] = split8(lds[lane + 64 * b], lds[lane + 64 * b + 1]);
    Parts x0 = split8(lds[lane], lds[lane + 7]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int base = (it & 3) * 512 + ks * 256;
            if (MODE == 1) {
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    w[b].h = __builtin_bit_cast(u32x4, lds[base + lane + 64 * b]);
                    w[b].m = __builtin_bit_cast(u32x4, lds[base + 1024 + lane + 64 * b]);
                    w[b].l = __builtin_bit_cast(u32x4, lds[base + 2048 + lane + 64 * b]);
                }
            } else if (MODE == 2) {
#pragma unroll
                for (int b = 0; b < TN; ++b) w[b] = split8(lds[base + 2 * lane + 128 * b], lds[base + 2 * lane + 128 * b + 1]);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                Parts x = x0;
                if (MODE != 0) x = split8(lds[base + 2048 + 2 * lane + 128 * a], lds[base + 2048 + 2 * lane + 128 * a + 1]);
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    f32x16 c = acc[a][b];
                    c = mfma32(w[b].l, x.h, c);
                    c = mfma32(w[b].m, x.m, c);
                    c = mfma32(w[b].h, x.l, c);
                    c = mfma32(w[b].m, x.h, c);
                    c = mfma32(w[b].h, x.m, c);
                    c = mfma32(w[b].h, x.h, c);
                    acc[a][b] = c;
                }
            }
        }
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r & 3] += acc[a][b][r];
    *reinterpret_cast<f32x4 *>(out + ((size_t)blockIdx.x * 256 + tid) * 4) = s;
}

template <int MODE, int TM, int TN>
static void run32(const char *name, float *out, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe32<MODE, TM, TN>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe32<MODE, TM, TN>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 32 * TM * TN * (double)iters * 4 * grid;
    printf("32x32x16 %-35s TM %d TN %d grid %4d  %8.3f ms  %7.1f TF fp32-equivalent  (%.1f TF bf16 MFMA)\n", name, TM, TN, grid,
           ms, flops / ms / 1e9, 6 * flops / ms / 1e9);
}

template <int MODE, int TM, int TN>
static void run(const char *name, float *out, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, TM, TN>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, TM, TN>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 32 * TM * TN * (double)iters * 4 * grid;      // fp32-equivalent
    printf("%-44s TM %d TN %d grid %4d  %8.3f ms  %7.1f TF fp32-equivalent  (%.1f TF bf16 MFMA)\n", name, TM, TN, grid, ms,
           flops / ms / 1e9, 6 * flops / ms / 1e9);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out;
    hipMalloc(&out, 1024 * 256 * 4 * sizeof(float));
    for (int grid : {256, 512}) {
        run<0, 4, 4>("mode 0: bf16 operands in registers", out, grid, iters);
        run<1, 4, 4>("mode 1: W parts from LDS, A split in wave", out, grid, iters);
        run<2, 4, 4>("mode 2: A and W split in wave", out, grid, iters);
        run<1, 4, 8>("mode 1: W parts from LDS, A split in wave", out, grid, iters);
        run<1, 2, 8>("mode 1: W parts from LDS, A split in wave", out, grid, iters);
        run<2, 4, 8>("mode 2: A and W split in wave", out, grid, iters);
        run32<0, 2, 2>("mode 0", out, grid, iters);
        run32<1, 2, 2>("mode 1: W parts, A split", out, grid, iters);
        run32<2, 2, 2>("mode 2: both split", out, grid, iters);
        run32<1, 2, 4>("mode 1: W parts, A split", out, grid, iters);
        run32<2, 2, 4>("mode 2: both split", out, grid, iters);
        run32<2, 1, 4>("mode 2: both split", out, grid, iters);
    }
    hipFree(out);
    return 0;
}
