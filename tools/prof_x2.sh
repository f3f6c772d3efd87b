cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PDGN_GEMM=x2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_x2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-eval-c5 > gpurun_out/kt_x2.log 2>&1
python3 tools/step_kernels.py gpurun_out/kt_x2 8 > gpurun_out/x2_step_kernels.txt 2>&1
python3 tools/x3_launches.py gpurun_out/kt_x2 6 20 > gpurun_out/x2_x3_launches.txt 2>&1
python3 tools/scan_launches.py gpurun_out/kt_x2 6 > gpurun_out/x2_scan_launches.txt 2>&1
head -40 gpurun_out/x2_step_kernels.txt
grep "^{" gpurun_out/kt_x2.log | cut -c1-300
rm -rf gpurun_out/kt_x2
