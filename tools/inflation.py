#!/usr/bin/env python3
"""Which kernels lose most to running next to others?  From a rocprofv3 --kernel-trace of tools/prof_list.py: per (kernel, grid) the
per-iteration time, the mean and the fastest instance, sorted by (mean - fastest) x launches -- the column-sum kernel's 1024-deep
atomic chains showed up here as 110 us mean against 15 us fastest.     usage: inflation.py <trace dir> [steps=6] [top=30]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
sel = rows[starts[-2 * steps - 1]:starts[-1]]
d = defaultdict(list)
for r in sel:
    n = r["Kernel_Name"].replace("void ", "").split("(")[0][:70]
    key = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Queue_Id"])
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for k, v in d.items():
    mean, mn = sum(v) / len(v), min(v)
    out.append(((mean - mn) * len(v) / steps, len(v) / steps, mean, mn, k))
out.sort(reverse=True)
print("lost us/step | launches/step | mean us | fastest us | queue | kernel (grid)")
for lost, cnt, mean, mn, k in out[:top]:
    print("%9.1f  %5.1f  %8.1f  %8.1f   q%s  %s (%s,%s,%s)" % (lost, cnt, mean, mn, k[4], k[0], k[1], k[2], k[3]))
