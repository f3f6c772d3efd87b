#!/usr/bin/env python3
"""The issuing queue's launches of ONE iteration in order, cut into ranges of launch indices, each range summed by kernel name:
what the small levels of a generator pass are made of.
usage: level_kernels.py <trace dir> "0:140,180:300,410:640" [top=14]"""
import csv, glob, os, sys
from collections import Counter, defaultdict
root = sys.argv[1]
ranges = [tuple(int(v) for v in r.split(":")) for r in sys.argv[2].split(",")]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
lo, hi = marks[-5], marks[-3]                              # one iteration: two generator passes
sel = rows[lo:hi]
byq = defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
mainq = max(byq, key=lambda q: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in byq[q]))
m = byq[mainq]
t0 = int(m[0]["Start_Timestamp"])
print("issuing queue %s: %d launches, %.2f ms from first start to last end" % (mainq, len(m), (int(m[-1]["End_Timestamp"]) - t0) / 1e6))
for a, b in ranges:
    part = m[a:min(b, len(m))]
    if not part:
        continue
    c, t = Counter(), Counter()
    for r in part:
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:80]
        c[n] += 1
        t[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(part[-1]["End_Timestamp"]) - int(part[0]["Start_Timestamp"])) / 1e3
    busy = sum(t.values())
    print("launches %d..%d: span %.0f us, busy %.0f us (%.1f us per launch), idle %.0f us" % (a, a + len(part), span, busy, busy / len(part), span - busy))
    for n, v in t.most_common(top):
        print("   %3d x %8.1f us  %s" % (c[n], v, n))
