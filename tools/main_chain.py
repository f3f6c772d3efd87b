#!/usr/bin/env python3
"""Where does the ISSUING stream's queue spend an iteration?  From a rocprofv3 --kernel-trace of tools/prof_list.py (or
bench.py): per hardware queue the busy time per step, and for the busiest queue (the issuing stream's) its kernels by name
and the idle gaps between them.
usage: main_chain.py <trace dir> [steps=6]"""
import csv, glob, os, sys
from collections import Counter, defaultdict
root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
first, last = starts[-2 * steps - 1], starts[-1]
sel = rows[first:last]
t_first, t_last = int(rows[first]["Start_Timestamp"]), int(rows[last]["Start_Timestamp"])
print("steps %d  wall %.2f ms/step  kernels/step %.0f" % (steps, (t_last - t_first) / 1e6 / steps, len(sel) / steps))
byq = defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
busy = {q: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v) / 1e6 / steps for q, v in byq.items()}
for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print("  queue %-4s %5d launches/step  busy %.2f ms/step" % (q, len(byq[q]) / steps, b))
mainq = max(busy, key=busy.get)
mr = byq[mainq]
c, t = Counter(), Counter()
gap_small = gap_big = 0.0
gaps = []
for a, b in zip(mr, mr[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g > 0:
        if g > 20:
            gap_big += g
            gaps.append((g, a["Kernel_Name"][:50], b["Kernel_Name"][:50]))
        else:
            gap_small += g
for r in mr:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:90]
    c[n] += 1
    t[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("issuing queue %s: busy %.2f ms/step, gaps <= 20 us: %.2f ms/step, gaps > 20 us: %.2f ms/step" % (
    mainq, busy[mainq], gap_small / 1e3 / steps, gap_big / 1e3 / steps))
for n, v in t.most_common(45):
    print("  %6.1f x %9.1f us/step  %s" % (c[n] / steps, v / steps, n))
gaps.sort(reverse=True)
print("largest gaps on the issuing queue (us, after, before):")
for g, a, b in gaps[:3 * steps:max(1, steps // 2)]:
    print("   %8.1f  %s -> %s" % (g, a, b))
