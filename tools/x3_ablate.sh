#!/bin/bash
# Run the ablation builds of tools/x3_bench.hip (tools/bin/x3b_<mask>, built on the CPU box:
#   for a in 0 1 2 4 8 16 32 17 19 27 31; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DX3_ABLATE=$a -mllvm -pragma-unroll-threshold=200000 -Ipdgn_amd/csrc tools/x3_bench.hip pdgn_amd/csrc/gemm_x3_16.hip pdgn_amd/csrc/gemm_x3_h2.hip -o tools/bin/x3b_$a; done)
# mask bits: 1 no conversion tasks, 2 no operand loads, 4 no stores, 8 no barrier, 16 no fragment reads, 32 no split arithmetic
out=${1:-gpurun_out/x3_ablate.txt}
mkdir -p $(dirname $out)
: > $out
for rep in 1 2; do
for a in 0 1 2 4 8 16 32 17 19 27 31; do
  [ -x tools/bin/x3b_$a ] && PDGN_NT_CFG=0 timeout 120 tools/bin/x3b_$a >> $out 2>&1
done
done
cat $out
