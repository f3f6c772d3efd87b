#!/bin/bash
# SQ counters of gemm_x3_kernel on the x3_bench shapes (GPU box, repo root): bash tools/x3_pmc.sh gpurun_out/x3pmc
set -u
OUT=${1:-gpurun_out/x3pmc}
mkdir -p $OUT
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -pragma-unroll-threshold=200000 -Ipdgn_amd/csrc tools/x3_bench.hip pdgn_amd/csrc/gemm_x3_16.hip pdgn_amd/csrc/gemm_x3_h2.hip pdgn_amd/csrc/gemm_rp.hip pdgn_amd/csrc/split.hip -o /tmp/x3b 2>/dev/null
export PDGN_NT_CFG=${PDGN_NT_CFG:-0}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- /tmp/x3b > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/b -- /tmp/x3b > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $OUT/c -- /tmp/x3b > $OUT/c.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
for sub in "abc":
    rows = []
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    by = {}
    for r in rows:
        if "gemm_x3" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"][:60], r["Grid_Size"])
        by.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        by[key].setdefault("_us", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for key, c in sorted(by.items()):
        print(sub, key, " ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(c.items())))
PY
