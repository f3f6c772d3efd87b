#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc counter_collection.csv.  usage: pmc_summary.py <dir> [name filter]"""
import csv, glob, sys, collections
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
    if flt and flt not in name:
        continue
    key = (name, r["Grid_Size"])
    d = agg.setdefault(key, collections.defaultdict(list))
    d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d["_dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), d in agg.items():
    n = len(d["_dur"])
    line = "%-58s grid %-9s" % (name, grid)
    for k, v in d.items():
        if k != "_dur":
            line += " %s=%.4g" % (k, sum(v) / len(v))
    print(line)
