#!/usr/bin/env python3
"""The dominant kernel's launches INSIDE the iteration, from a rocprofv3 --kernel-trace of bench.py: conv2's dense half at stage 4,
forward = the pre-split-weight instance of gemm_x3_kernel at the grid pdgn_gemm_nt_ps gives that problem (512 workgroups of 256 at
B = 35).  Prints their average duration: the figure bench.py's `roofline.us_per_launch` (HIP events around the same launches inside
the timed steps) must agree with.   usage: conv2_in_step.py <trace dir> [grid_threads=131072]"""
import csv, glob, os, sys
root = sys.argv[1]
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
allrows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
DP, SK = "gemm_x3_kernel<4, 2, 2, 2, 1, false, false, false, false, true", "gemm_x3_kernel<4, 2, 2, 2, 1, true, false, false, false, true"
d = []
for i, r in enumerate(allrows):
    if DP in r["Kernel_Name"] and int(r["Grid_Size_X"]) == grid:
        # the call = this data-parallel launch + the stream-K tail launch that follows it on the same queue (560 tiles: 512 + 48)
        t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        for q in allrows[i + 1:i + 40]:
            if SK in q["Kernel_Name"] and q["Queue_Id"] == r["Queue_Id"]:
                t1 = int(q["End_Timestamp"])
                break
        d.append((t1 - t0) / 1e3)
if not d:
    print("no launch of that instance with grid", grid)
    sys.exit(1)
warm = d[len(d) // 3:]                       # skip the warm-up / capture iterations at the head of the trace
flops = 2.0 * 35840 * 512 * 5120
print("%d calls (data-parallel launch at grid %d + its stream-K tail, first start to last end) in the trace; the last %d: mean %.1f us  min %.1f  max %.1f  ->  %.1f TFLOP/s = %.3f of 416.7"
      % (len(d), grid, len(warm), sum(warm) / len(warm), min(warm), max(warm), flops / (sum(warm) / len(warm)) / 1e6,
         flops / (sum(warm) / len(warm)) / 1e6 / 416.7))
