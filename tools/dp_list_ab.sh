#!/bin/bash
# PDGN_FORCE_DIST=0/1 (single process / one-rank RCCL group), launch list in both: ms per step, alternating, one box.
out=${1:-gpurun_out/dp_list_ab.txt}; n=${2:-3}
: > $out
for rep in $(seq $n); do for m in 0 1; do
  PDGN_FORCE_DIST=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-eval-c5 --no-roofline > /tmp/b.log 2>&1
  rc=$?
  line=$(grep "^{" /tmp/b.log | tail -1)
  if [ -z "$line" ]; then echo "force_dist $m rc $rc NO LINE" | tee -a $out; tail -25 /tmp/b.log | tee -a $out
  else echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force_dist $m rc $rc', round(d['ms_per_step'],3), d['config']['issue'], d['rccl_ranks'])" | tee -a $out; fi
done; done
