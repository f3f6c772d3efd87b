// common.h -- shared helpers for the gfx950 kernels of libpdgn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pdgn_hip.h"

#define PDGN_WAVE 64

static inline int pdgn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Squared distance exactly as the reference kernels compute it after nvcc's FMA
// contraction (knnquery_cuda_kernel.cu:31, interpolation_cuda_kernel.cu:156):
// fma(dz,dz, fma(dy,dy, dx*dx)) with d = q - p.  The CPU oracle spells the same chain.
__device__ __forceinline__ float sqdist3(float qx, float qy, float qz, float px, float py, float pz) {
    float dx = qx - px, dy = qy - py, dz = qz - pz;
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (PDGN_WAVE - 1); }
