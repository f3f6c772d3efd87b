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
# an iteration = two generator passes; it begins at the mark that follows the optimizer's multi-tensor launch of the previous one
def _starts_iteration(j):                                  # marks[j]: is the generator's optimizer launch between marks[j - 1] and it, on its queue?
    q = rows[marks[j]]["Queue_Id"]
    return any(r["Queue_Id"] == q and "multi_tensor_apply" in r["Kernel_Name"] and "FusedOptimizer" in r["Kernel_Name"]
               for r in rows[marks[j - 1]:marks[j]])
first = next(j for j in range(len(marks) - 5, 0, -1) if _starts_iteration(j))
lo, hi = marks[first], marks[first + 2]                              # one iteration: two generator passes
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
