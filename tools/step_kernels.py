#!/usr/bin/env python3
"""Per-step kernel time by kernel name from a rocprofv3 --kernel-trace of bench.py (last N steps averaged).
usage: step_kernels.py <trace dir> [steps=10]"""
import csv, glob, os, sys
from collections import Counter
root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step = from one fc1 small_mlp_fwd of generator pass #1 to the next but one (two passes per step)
starts = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
first, last = starts[-2 * steps - 1], starts[-1]
sel = rows[first:last]
wall = (int(rows[last]["Start_Timestamp"]) - int(rows[first]["Start_Timestamp"])) / 1e6 / steps
c, t = Counter(), Counter()
for r in sel:
    n = r["Kernel_Name"].replace("void ", "").split("(")[0][:70]
    c[n] += 1
    t[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("steps %d  wall %.2f ms/step  kernels/step %.0f  sum of kernel time %.2f ms/step" % (steps, wall, len(sel) / steps, sum(t.values()) / 1e3 / steps))
groups = {"gemm_x3": 0.0, "gemm_nt": 0.0, "gemm_tn": 0.0, "Cijk": 0.0, "at::native": 0.0, "wgs": 0.0, "cl_": 0.0, "bn_softmax|bilateral": 0.0, "knn": 0.0}
for n, v in t.items():
    for g in groups:
        if any(x in n for x in g.split("|")):
            groups[g] += v
            break
print("  " + "  ".join("%s %.2f" % (g, v / 1e3 / steps) for g, v in groups.items()))
for n, v in t.most_common(70):
    print("  %6.1f x %9.1f us/step  %s" % (c[n] / steps, v / steps, n))
