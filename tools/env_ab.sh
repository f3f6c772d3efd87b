#!/bin/bash
# Generic in-step A/B of one environment switch: tools/env_ab.sh VAR "v0 v1 ..." [reps] -- bench.py per value, alternating, one box.
var=$1; vals=$2; reps=${3:-3}
for rep in $(seq $reps); do for v in $vals; do
  ms=$(env $var=$v timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-eval-c5 --no-roofline 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$var=$v  $ms ms/step"
done; done
