#!/usr/bin/env python3
"""The dominant kernel's launches INSIDE the iteration, from a rocprofv3 --kernel-trace of bench.py: conv2's dense half at stage 4,
forward = the pre-split-weight instance of gemm_x3_kernel at the grid pdgn_gemm_nt_ps gives that problem (512 workgroups of 256 at
B = 35), with the stream-K tail and its reduce behind it.  Prints their average duration: the figure bench.py's `roofline.us_per_launch` (HIP events around the same launches inside
the timed steps) must agree with.   usage: conv2_in_step.py <trace dir> [grid_threads=131072]"""
import csv, glob, os, sys
root = sys.argv[1]
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 512            # workgroups of the data-parallel launch
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
allrows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
# plain pre-split instance of the 256 x 128 tile: four waves of 128 x 64 or (two parts) eight of 64 x 64
PWS = ("gemm_x3_kernel<4, 2, 2, 2, 1, false, false, false, false, true", "gemm_x3_kernel<2, 2, 4, 2, 1, false, false, false, false, true")
wgs = lambda r: int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
d, parts = [], 3
for i, r in enumerate(allrows):
    if any(p in r["Kernel_Name"] for p in PWS) and wgs(r) == grid:
        # the call = this data-parallel launch + what follows it on the same queue: the stream-K tail (the same instance over the 48
        # leftover tiles' k slices) and the reduce of its partial tiles (560 tiles: 512 + 48); the atomic form has no reduce
        parts = 2 if r["Kernel_Name"].rstrip(">(NtArgs) ").endswith(", 2") or ", 32, 2>" in r["Kernel_Name"] else 3
        t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nxt = [q for q in allrows[i + 1:i + 60] if q["Queue_Id"] == r["Queue_Id"]][:2]
        if nxt and any(p in nxt[0]["Kernel_Name"] for p in PWS) and wgs(nxt[0]) != grid:
            t1 = int(nxt[0]["End_Timestamp"])
            if len(nxt) > 1 and "x3_sk_reduce" in nxt[1]["Kernel_Name"]:
                t1 = int(nxt[1]["End_Timestamp"])
        d.append((t1 - t0) / 1e3)
if not d:
    print("no launch of that instance with grid", grid)
    sys.exit(1)
warm = d[len(d) // 3:]                       # skip the warm-up / capture iterations at the head of the trace
flops = 2.0 * 35840 * 512 * 5120
roof = 2500.0 / (3 if parts == 2 else 6)
print("%d calls (data-parallel launch of %d workgroups + stream-K tail + reduce, first start to last end) in the trace; the last %d: mean %.1f us  min %.1f  "
      "max %.1f  ->  %.1f TFLOP/s = %.3f of %.1f (%d parts: %d matrix-core products per fp32 product)"
      % (len(d), grid, len(warm), sum(warm) / len(warm), min(warm), max(warm), flops / (sum(warm) / len(warm)) / 1e6,
         flops / (sum(warm) / len(warm)) / 1e6 / roof, roof, parts, 3 if parts == 2 else 6))
