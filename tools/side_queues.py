#!/usr/bin/env python3
"""Kernel time per iteration of the SIDE queues (everything but the issuing stream's), by kernel name, from a rocprofv3
--kernel-trace of tools/prof_list.py: what the discriminators' / the losses' streams spend, which is what the issuing stream's
kernels share the chip with.   usage: side_queues.py <trace dir> [steps=6] [top=22]"""
import csv, glob, os, sys
from collections import Counter, defaultdict
root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
top = int(sys.argv[3]) if len(sys.argv) > 3 else 22
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
first, last = starts[-2 * steps - 1], starts[-1]
sel = rows[first:last]
byq = defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
busy = {q: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v) / 1e6 / steps for q, v in byq.items()}
mainq = max(busy, key=busy.get)
print("wall %.2f ms/step" % ((int(rows[last]["Start_Timestamp"]) - int(rows[first]["Start_Timestamp"])) / 1e6 / steps))
for q in sorted(busy, key=lambda q: -busy[q]):
    print("queue %-3s %5.0f launches/step  busy %6.2f ms/step%s" % (q, len(byq[q]) / steps, busy[q], "  (issuing)" if q == mainq else ""))
for q in sorted(busy, key=lambda q: -busy[q]):
    if q == mainq:
        continue
    c, t = Counter(), Counter()
    for r in byq[q]:
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:80]
        c[n] += 1
        t[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("queue %s:" % q)
    for n, v in t.most_common(top):
        print("  %6.1f x %9.1f us/step  %s" % (c[n] / steps, v / steps, n))
