// x3_inst.hip -- ONE instance of gemm_x3_kernel for ISA / register inspection (the library's translation unit takes minutes):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -pragma-unroll-threshold=200000 -Ipdgn_amd/csrc \
//         -DX3_INST="4,2,2,2,1,false,false,false,false,true,16" -S --cuda-device-only tools/x3_inst.hip -o /tmp/x3_inst.s
#define X3_KERNEL_ONLY
#include "../pdgn_amd/csrc/gemm_x3.hip"
#ifndef X3_INST
#define X3_INST 4, 2, 2, 2, 1, false, false, false, false, true, 16
#endif
template __global__ void gemm_x3_kernel<X3_INST>(const NtArgs);
