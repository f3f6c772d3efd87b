#!/bin/bash
# A/B of this tree against the copy under _prev/ (a built checkout of an earlier commit) on one box, alternating:
#   (here)  rm -rf _prev && mkdir _prev && git archive <commit> | tar -x -C _prev && (cd _prev && python -m pdgn_amd.build)
#   gpurun -- 'bash tools/ab_trees.sh 3'
N=${1:-3}
cp tools/step_time.py _prev/tools/step_time.py
for i in $(seq $N); do
  python3 tools/step_time.py 30 2>&1 | grep "ms/step"
  (cd _prev && python3 tools/step_time.py 30 2>&1 | grep "ms/step")
done
