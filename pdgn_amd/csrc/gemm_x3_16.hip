// gemm_x3_16.hip -- the v_mfma_f32_16x16x32_bf16 instances of gemm_x3_kernel (gemm_x3.hip, template parameter MS = 16) in a
// translation unit of their own: 33 instances of a 192-instruction straight-line chunk take as long to compile as the 33 of the
// default form, and the two files build side by side.  gemm_x3.hip's launchers reach them through x3_launch16.
#define X3_KERNEL_ONLY
#include "gemm_x3.hip"

template <int TM, int TN, int WM, int WN, int OCC>
static bool x3_launch16_tile(int flags, int grid, hipStream_t s, const NtArgs &a) {
#define X3_CASE(A_, WT_, AT_, EPI_, PW_)                                                                                          \
    case ((A_ ? 1 : 0) | (WT_ ? 2 : 0) | (AT_ ? 4 : 0) | (EPI_ ? 8 : 0) | (PW_ ? 16 : 0)):                                         \
        hipLaunchKernelGGL((gemm_x3_kernel<TM, TN, WM, WN, OCC, A_, WT_, AT_, EPI_, PW_, 16>), dim3(grid), dim3(64 * WM * WN), 0, s, a); \
        return true
    switch (flags) {                                            // exactly the combinations X3Cfg::launch can ask for
        X3_CASE(true, false, false, false, false);
        X3_CASE(false, false, false, true, true);
        X3_CASE(false, false, false, false, true);
        X3_CASE(true, false, false, false, true);
        X3_CASE(false, false, false, true, false);
        X3_CASE(false, false, false, false, false);
        X3_CASE(true, true, false, false, false);
        X3_CASE(false, true, false, true, false);
        X3_CASE(false, true, false, false, false);
        X3_CASE(true, true, true, false, false);
        X3_CASE(false, true, true, false, false);
        default: return false;
    }
#undef X3_CASE
}

void x3_launch16(int cfg, int flags, int grid, hipStream_t s, const NtArgs &a) {
    bool ok;
    if (cfg == 0) ok = x3_launch16_tile<4, 2, 2, 2, 1>(flags, grid, s, a);          // X3Big
    else if (cfg == 1) ok = x3_launch16_tile<2, 2, 2, 2, 1>(flags, grid, s, a);     // X3Square
    else ok = x3_launch16_tile<2, 1, 2, 2, 2>(flags, grid, s, a);                   // X3Narrow
    if (!ok) abort();                                           // a launcher asked for an instance that does not exist: a build error, not a run-time condition
}
