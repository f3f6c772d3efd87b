#!/bin/bash
# In-step A/B of the matrix instruction per instance class of gemm_x3 (PDGN_X3_SHAPE16_MASK, bit 4 * tile + class):
# one bench.py run per mask, alternating, on ONE box.  usage: tools/shape_mask_ab.sh out.txt mask mask ...
out=$1; shift
: > $out
for rep in 1 2; do
for m in "$@"; do
  ms=$(PDGN_X3_SHAPE16_MASK=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-eval-c5 --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "mask $m  $ms ms/step" | tee -a $out
done
done
