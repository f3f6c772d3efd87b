#!/usr/bin/env python3
"""Every gemm_x3 / reduce launch on the issuing queue of one iteration, in order, from a rocprofv3 --kernel-trace of bench.py
(instance, workgroups, duration): what a contraction call is made of inside the step.   usage: x3_launches.py <trace dir> [steps=6] [min_us=0]"""
import csv, glob, os, re, sys, collections
root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
sel = rows[starts[-2 * steps - 1]:starts[-1]]
agg = collections.OrderedDict()
for r in sel:
    n = r["Kernel_Name"]
    m = re.search(r"gemm_x3_kernel<([^>]*)>", n)
    if not m and "x3_sk_reduce" not in n:
        continue
    key = (m.group(1) if m else "x3_sk_reduce_kernel", int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Queue_Id"])
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = collections.Counter()
for (inst, g, q), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("queue %s  %-58s grid %5d: %6.1f x %8.1f us = %8.1f us/step" % (q, "<" + inst + ">", g, len(v) / steps, sum(v) / len(v), sum(v) / steps))
    tot[q] += sum(v) / steps
print({q: round(t, 1) for q, t in tot.items()})
