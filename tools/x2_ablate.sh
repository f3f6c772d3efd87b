cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
: > gpurun_out/x2_ablate.txt
for rep in 1 2; do for a in 0 1 2 3 16 32 31; do PDGN_NT_CFG=0 timeout 120 tools/bin/x3b_$a >> gpurun_out/x2_ablate.txt 2>&1; done; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-eval-c5 > gpurun_out/kt_b.log 2>&1
python3 tools/conv2_in_step.py gpurun_out/kt_b > gpurun_out/r05_conv2_in_step.txt 2>&1
rm -rf gpurun_out/kt_b
cat gpurun_out/x2_ablate.txt | cut -c1-250; cat gpurun_out/r05_conv2_in_step.txt
