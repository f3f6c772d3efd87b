#!/usr/bin/env python3
"""Stream-K tail launches (ATOMIC, non-transposed instances of gemm_x3_kernel) of one iteration from a rocprofv3 --kernel-trace of
bench.py: grid, duration, and the data-parallel launch in front of each.   usage: sk_tails.py <trace dir> [steps=6]"""
import csv, glob, os, sys
root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
first, last = starts[-2 * steps - 1], starts[-1]
sel = rows[first:last]
import collections, re
agg = collections.OrderedDict()
for i, r in enumerate(sel):
    m = re.search(r"gemm_x3_kernel<([^>]*)>", r["Kernel_Name"])
    if not m:
        continue
    a = [t.strip() for t in m.group(1).split(",")]
    if a[5] != "true" or a[7] == "true":          # ATOMIC and not AT (weight gradients are split-K launches of their own)
        continue
    prev = next((q for q in reversed(sel[:i]) if q["Queue_Id"] == r["Queue_Id"] and "gemm_x3_kernel" in q["Kernel_Name"]), None)
    key = (m.group(1), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(prev["Grid_Size_X"]) // int(prev["Workgroup_Size_X"]) if prev else -1, r["Queue_Id"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    pd = (int(prev["End_Timestamp"]) - int(prev["Start_Timestamp"])) / 1e3 if prev else 0.0
    agg.setdefault(key, []).append((d, pd))
tot = 0.0
for (inst, g, pg, q), v in agg.items():
    d = sum(x for x, _ in v) / len(v)
    pd = sum(y for _, y in v) / len(v)
    tot += sum(x for x, _ in v) / steps
    print("queue %s  <%s>  tail grid %4d: %7.1f us   (x %.1f per step)   behind a launch of grid %5d: %7.1f us" % (q, inst, g, d, len(v) / steps, pg, pd))
print("stream-K tails: %.1f us per step" % tot)
