#!/bin/bash
# One kernel trace of the bench step and its reductions into gpurun_out/ (GPU box, repo root): bash tools/step_trace.sh [tag]
TAG=${1:-now}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$TAG -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-eval-c5 > gpurun_out/kt_$TAG.log 2>&1
python3 tools/step_kernels.py gpurun_out/kt_$TAG 8 > gpurun_out/${TAG}_step_kernels.txt 2>&1
python3 tools/x3_launches.py gpurun_out/kt_$TAG 6 20 > gpurun_out/${TAG}_x3_launches.txt 2>&1
python3 tools/scan_launches.py gpurun_out/kt_$TAG 6 > gpurun_out/${TAG}_scan_launches.txt 2>&1
python3 tools/main_chain.py gpurun_out/kt_$TAG 6 > gpurun_out/${TAG}_main_chain.txt 2>&1
python3 tools/conv2_in_step.py gpurun_out/kt_$TAG > gpurun_out/${TAG}_conv2_in_step.txt 2>&1
[ -n "$KEEP_TRACE" ] || rm -rf gpurun_out/kt_$TAG
