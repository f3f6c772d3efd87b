// gemm_x3_h2.hip -- the two-part instances of gemm_x3_kernel (gemm_x3.hip, template parameter NP = 2: two scaled fp16 parts per
// operand value, three partial products on v_mfma_f32_32x32x16_f16) in a translation unit of their own, built beside the bf16 forms:
// the 256 x 128 tile only (x2_pays: the launches whose time is their matrix-core work).  gemm_x3.hip's launchers reach them through
// x3_launch_h2.
#define X3_KERNEL_ONLY
#include "gemm_x3.hip"

// The 256 x 128 tile in two shapes.  Eight waves of 64 x 64 (two per SIMD; 174 vector registers): while one wave's matrix
// instructions run the other issues its split / LDS / load instructions -- the two-part loop has half the matrix-core work of the
// bf16 form to hide the same operand path behind, and one wave per SIMD no longer hides it (alone: conv2's dense half 950 -> 860 us,
// the per-point product 750 -> 715, 71680 x 1024 x 256 275 -> 253; identical results).  Four waves of 128 x 64 for the launches
// that emit BatchNorm partials: their layout (two partial rows per tile row, pdgn_gemm_nt_stat_rows) is the four-wave tile's.
// exactly the flag combinations X3Cfg::launch can ask for (flags = ATOMIC | WT << 1 | AT << 2 | EPI << 3 | PW << 4)
#define X2_INSTANCES(X)                                                                                                          \
    X(true, false, false, false, false) X(false, false, false, true, true) X(false, false, false, false, true)                    \
    X(true, false, false, false, true) X(false, false, false, true, false) X(false, false, false, false, false)                   \
    X(true, true, false, false, false) X(false, true, false, true, false) X(false, true, false, false, false)                     \
    X(true, true, true, false, false) X(false, true, true, false, false)

template <int TM, int TN, int WM, int WN>
static const void *x2_symbol_tile(int flags) {
#define X2_CASE(A_, WT_, AT_, EPI_, PW_)                                                                                          \
    case ((A_ ? 1 : 0) | (WT_ ? 2 : 0) | (AT_ ? 4 : 0) | (EPI_ ? 8 : 0) | (PW_ ? 16 : 0)):                                         \
        return (const void *)gemm_x3_kernel<TM, TN, WM, WN, 1, A_, WT_, AT_, EPI_, PW_, 32, 2>;
    switch (flags) {
        X2_INSTANCES(X2_CASE)
        default: return nullptr;
    }
#undef X2_CASE
}

const void *x3_symbol_h2(int flags, bool eight_waves) {
    return eight_waves ? x2_symbol_tile<2, 2, 4, 2>(flags) : x2_symbol_tile<4, 2, 2, 2>(flags);
}

void x3_launch_h2(int flags, int grid, hipStream_t s, const NtArgs &a, bool eight_waves) {
    const void *f = x3_symbol_h2(flags, eight_waves);
    if (!f) abort();                                            // a launcher asked for an instance that does not exist: a build error, not a run-time condition
    NtArgs args = a;
    void *params[] = {&args};
    (void)hipLaunchKernel(f, dim3(grid), dim3(eight_waves ? 512 : 256), params, 0, s);
}
