#!/bin/bash
# Per-entry rocprofv3 --pmc passes over tools/roofline_entry.py (separate passes: SQ/GRBM counters, FETCH_SIZE, WRITE_SIZE).
# usage (GPU box, from the repo root): bash tools/run_pmc_roofline.sh gpurun_out/pmc_r03
set -u
OUT=${1:-gpurun_out/pmc_r03}
for e in conv2_dense_stage4 per_point_stage4 conv2_dense_dx_stage4 weight_grad_stage4 bn_act_backward_stage4 window_gather_sum_stage4 feature_knn_stage4 knn3_largest; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/$e/mfma -- python3 tools/roofline_entry.py $e > $OUT.$e.mfma.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$e/fetch -- python3 tools/roofline_entry.py $e > $OUT.$e.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$e/write -- python3 tools/roofline_entry.py $e > $OUT.$e.write.log 2>&1
done
# config C5's kernel (the evaluation path): vector-ALU issue counters only
rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/emd_cost_c5/valu -- python3 tools/roofline_entry.py emd_cost_c5 > $OUT.emd.log 2>&1
