#!/usr/bin/env python3
"""x2_maxima_kernel launches (with the kernel in front of and behind each on its queue) of one iteration from a rocprofv3 --kernel-trace of bench.py (PDGN_GEMM=x2): by grid, count, duration,
and the kernel that follows each on the same queue.   usage: scan_launches.py <trace dir> [steps=6]"""
import csv, glob, os, sys, collections
root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
sel = rows[starts[-2 * steps - 1]:starts[-1]]
agg = collections.OrderedDict()
for i, r in enumerate(sel):
    if "x2_maxima" not in r["Kernel_Name"]:
        continue
    nxt = next((q for q in sel[i + 1:i + 40] if q["Queue_Id"] == r["Queue_Id"] and "x2_maxima" not in q["Kernel_Name"]), None)
    prv = next((q for q in reversed(sel[max(0, i - 40):i]) if q["Queue_Id"] == r["Queue_Id"] and "x2_maxima" not in q["Kernel_Name"]), None)
    short = lambda q: (q["Kernel_Name"].replace("void ", "").split("(")[0][:44] if q else "-")
    key = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Queue_Id"], short(prv) + "  ->  " + short(nxt))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for (g, q, n), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v) / steps
    print("queue %s grid %5d: %5.1f x %8.1f us = %8.1f us/step   -> %s" % (q, g, len(v) / steps, sum(v) / len(v), sum(v) / steps, n))
print("total %.1f us/step" % tot)
