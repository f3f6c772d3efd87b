#!/usr/bin/env python3
"""The largest torch-glue launches (at::native / rocclr fill+copy) of ONE step of a rocprofv3 --kernel-trace of bench.py,
with grid size, queue and the own kernels launched before / after on the same queue (to find them in the host code).
usage: glue_top.py <trace dir> [top=40]"""
import csv, glob, os, sys
root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
first, last = starts[-3], starts[-1]
sel = rows[first:last]
t0 = int(sel[0]["Start_Timestamp"])


def short(n):
    n = n.replace("void ", "").replace("at::native::", "")
    for a, b in (("vectorized_elementwise_kernel", "vec"), ("elementwise_kernel_manual_unroll", "elt"), ("(anonymous namespace)::", "")):
        n = n.replace(a, b)
    return n.split("(")[0][:64] if not n.startswith(("vec", "elt")) else n[:90]


def glue(r):
    n = r["Kernel_Name"]
    return "at::native" in n or "rocclr" in n


byq = {}
for i, r in enumerate(sel):
    byq.setdefault(r["Queue_Id"], []).append(i)
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel if glue(r)) / 1e3
print("one step: %d launches, %d glue launches, %.0f us of glue kernel time" % (len(sel), sum(map(glue, sel)), tot))
big = sorted((i for i, r in enumerate(sel) if glue(r)), key=lambda i: int(sel[i]["Start_Timestamp"]) - int(sel[i]["End_Timestamp"]))[:top]
for i in sorted(big):
    r = sel[i]
    q = byq[r["Queue_Id"]]
    p = q.index(i)
    prev = next((short(sel[j]["Kernel_Name"]) for j in reversed(q[:p]) if not glue(sel[j])), "-")
    nxt = next((short(sel[j]["Kernel_Name"]) for j in q[p + 1:] if not glue(sel[j])), "-")
    print("%8.2f ms  %7.1f us  grid %10d  q%-3s %-70s  after %-34s before %s" % (
        (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
        int(r["Grid_Size_X"]), r["Queue_Id"], short(r["Kernel_Name"]), prev[:34], nxt[:40]))
