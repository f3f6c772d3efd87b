#!/usr/bin/env python3
"""What runs while the issuing stream WAITS?  From a rocprofv3 --kernel-trace of tools/prof_list.py: the longest idle gap of the
busiest hardware queue in one of the last iterations, and every kernel of the other queues that overlaps it (start / end
relative to the gap's start, in us).
usage: wait_gap.py <trace dir> [iteration from the end = 2]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
per = 2                                                   # the generator's noise MLP runs twice an iteration
lo, hi = marks[-per * back - 1], marks[-per * (back - 1) - 1]
sel = rows[lo:hi]
byq = defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
mainq = max(byq, key=lambda q: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in byq[q]))
m = byq[mainq]
gaps = [(int(m[i + 1]["Start_Timestamp"]) - int(m[i]["End_Timestamp"]), i) for i in range(len(m) - 1)]
gaps.sort(reverse=True)
for g, i in gaps[:2]:
    g0, g1 = int(m[i]["End_Timestamp"]), int(m[i + 1]["Start_Timestamp"])
    print("gap of %.0f us on queue %s after %s, before %s" % (g / 1e3, mainq, m[i]["Kernel_Name"][:50], m[i + 1]["Kernel_Name"][:50]))
    for q in sorted(byq):
        if q == mainq:
            continue
        for r in byq[q]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if e > g0 and s < g1:
                print("   q%s %8.1f .. %8.1f  (%6.1f us)  %s" % (q, (s - g0) / 1e3, (e - g0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
