#!/usr/bin/env python3
"""What runs beside a slow instance?  For every launch of <pattern> in the trace's last iterations that took longer than <min us>: the
kernels of ALL queues that overlap it (start / end relative to its start).
usage: overlap_of.py <trace dir> <pattern> <min us>"""
import csv, glob, os, sys
root, pat, mn = sys.argv[1], sys.argv[2], float(sys.argv[3])
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
hits = [r for r in rows if pat in r["Kernel_Name"] and (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 >= mn]
for h in hits[:3]:
    a, b = int(h["Start_Timestamp"]), int(h["End_Timestamp"])
    print("== %s on q%s: %.1f us (grid %s)" % (h["Kernel_Name"][:50], h["Queue_Id"], (b - a) / 1e3, h["Grid_Size_X"]))
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e > a - 30000 and s < b + 10000 and r is not h:
            print("   q%s  %8.1f .. %8.1f  (%7.1f us)  grid %8s  %s" % (r["Queue_Id"], (s - a) / 1e3, (e - a) / 1e3, (e - s) / 1e3, r["Grid_Size_X"],
                                                                    r["Kernel_Name"].replace("void ", "")[:70]))
