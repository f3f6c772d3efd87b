#!/bin/bash
# Everything under profiles/ for one round, on one MI355X box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh gpurun_out/prof r06
set -u
OUT=${1:-gpurun_out/prof}; TAG=${2:-r06}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# the bench line as the driver runs it (launch-list issue, cpu_baseline at B = 35), and the same with every launch issued from Python
python3 bench.py --steps 30 --warmup 5 > $OUT/${TAG}_bench_full.json 2> $OUT/bench.err
python3 bench.py --steps 30 --warmup 5 --issue eager --no-cpu-baseline --no-eval-c5 > $OUT/${TAG}_bench_eager.json 2>> $OUT/bench.err
python3 bench.py --eval --steps 10 --warmup 2 > $OUT/${TAG}_eval_c5.json 2>> $OUT/bench.err
# kernel trace + stats of the bench command (the launch-list step) and its reductions
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-eval-c5 > $OUT/kt_bench.log 2>&1
cp $(find $OUT/kt_bench -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
python3 tools/step_kernels.py $OUT/kt_bench 8 > $OUT/${TAG}_step_kernels.txt
python3 tools/main_chain.py $OUT/kt_bench 6 > $OUT/${TAG}_main_chain.txt
python3 tools/side_queues.py $OUT/kt_bench 6 22 > $OUT/${TAG}_side_queues.txt 2>&1
python3 tools/wait_gap.py $OUT/kt_bench 2 > $OUT/${TAG}_wait_gap.txt 2>&1
python3 tools/conv2_in_step.py $OUT/kt_bench > $OUT/${TAG}_conv2_in_step.txt 2>&1
python3 tools/x3_launches.py $OUT/kt_bench 6 20 > $OUT/${TAG}_x3_launches.txt 2>&1
python3 tools/scan_launches.py $OUT/kt_bench 6 > $OUT/${TAG}_scan_launches.txt 2>&1
MS=$(python3 -c "import json,sys; print(json.load(open(sys.argv[1]))[\"ms_per_step\"])" $OUT/${TAG}_bench_full.json)
python3 tools/queue_timeline.py $OUT/kt_bench $MS > $OUT/${TAG}_queue_timeline.txt 2>&1
# the roofline kernels alone
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_roof -- python3 tools/roofline_only.py > $OUT/${TAG}_roofline_only.json 2> $OUT/kt_roof.log
cp $(find $OUT/kt_roof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_roofline_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_eval -- python3 bench.py --eval --steps 5 --warmup 1 > $OUT/kt_eval.log 2>&1
cp $(find $OUT/kt_eval -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_eval_kernel_stats.csv
# counters (separate --pmc passes), traffic.json
bash tools/run_pmc_roofline.sh $OUT/pmc > /dev/null 2>&1
python3 tools/pmc_roofline.py $OUT/pmc $OUT/${TAG}_pmc_mfma.csv $OUT/traffic.json > $OUT/${TAG}_pmc_summary.txt
# every GEMM problem of a step alone; both contraction kernels against fp64; the pre-split second operand
ALL_CFGS=1 python3 tools/gemm_shapes.py > $OUT/${TAG}_gemm_shapes.txt 2>&1
python3 tools/x3_check.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_x3_check.txt
PDGN_GEMM=x3 python3 tools/x3_check.py 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_x3_check.txt
PDGN_GEMM=fp32 python3 tools/x3_check.py 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_x3_check.txt
PDGN_GEMM=x3 python3 tools/ps_check.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_presplit_check.txt
bash tools/x3_pmc.sh $OUT/x3pmc 2>&1 | grep -E "^[abc] \(" > $OUT/${TAG}_x3_pmc.txt
# where the iteration's time goes, untraced: progress of the issuing stream / D4's / the local-pair loss's chain through one list
for c in 0 4 5; do python3 -u tools/list_progress.py 10 $c 2>&1 | grep -v amdgpu.ids; done > $OUT/${TAG}_list_progress.txt
python3 tools/host_time.py > $OUT/${TAG}_host_time.txt 2>&1
python3 tools/phase_events.py > $OUT/${TAG}_phases_overlapped.txt 2>&1
python3 tools/finalize_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_finalize_bench.txt
python3 bench.py --base-points 256 --steps 10 --warmup 3 --no-cpu-baseline --no-eval-c5 > $OUT/${TAG}_bench_c4.json 2>> $OUT/bench.err
# round 5: both bf16 matrix instructions (x3_check with the shape forced), the launch list under a one-rank RCCL group, the
# torch-native / library launches with their owners, the block fixtures' backward error, the register table of the shipped library
PDGN_GEMM=x3 python3 tools/x3_check.py 16 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_x3_check_16x16x32.txt
bash tools/dp_list_ab.sh $OUT/${TAG}_dp_list_ab.txt 3 > /dev/null 2>&1
python3 tools/glue_owners.py 35 2>&1 | grep -v -E "amdgpu.ids|Warning|_warn" > $OUT/${TAG}_glue_owners.txt
python3 tools/backward_error.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_backward_error.txt
python3 tools/spill_table.py > $OUT/${TAG}_spill_table.txt 2>&1
# the arithmetic modes of the contractions: two fp16 parts where they pay (default) against three bf16 parts everywhere, alternating
# in the step; every contraction of an iteration alone in both; all three modes against fp64
bash tools/env_ab.sh PDGN_GEMM "x2 x3" 3 > $OUT/${TAG}_gemm_mode_ab.txt 2>&1
python3 tools/x2_shapes.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_x2_shapes.txt
python3 tools/x2_check.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_x2_check.txt
python3 tools/cfg_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_cfg_probe.txt
python3 tools/operand_range.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_operand_range.txt
# stream-K tails: workspace + reduce kernel (default) against the atomic form, alternating in the step
bash tools/env_ab.sh PDGN_X3_SK_WS "1 0" 3 > $OUT/${TAG}_sk_tails_ab.txt 2>&1
# round 6: the row-panel kernel of the short-reduction products alone (against the tile kernel; store policies) and in the step; the
# F = 128, N = 512 block of the imported reference in the three arithmetic modes
for rp in 1 0; do PDGN_RP=$rp python3 tools/rp_bench.py 2>&1 | grep -v amdgpu.ids; done > $OUT/${TAG}_rp_bench.txt
for st in 0 2 16; do PDGN_RP_STORE=$st python3 tools/rp_bench.py 2>&1 | grep -v amdgpu.ids; done >> $OUT/${TAG}_rp_bench.txt
PDGN_RP_PIPE=0 python3 tools/rp_bench.py 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_rp_bench.txt
bash tools/env_ab.sh PDGN_RP "1 0" 3 > $OUT/${TAG}_rp_ab.txt 2>&1
python3 tools/block_big_error.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_block_big_error.txt
rm -rf $OUT/kt_bench $OUT/kt_roof $OUT/kt_eval $OUT/pmc $OUT/pmc.*.log $OUT/x3pmc
ls -la $OUT
