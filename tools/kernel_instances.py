#!/usr/bin/env python3
"""Every launch of the kernels whose name contains <pattern> in the last full iteration of a kernel trace: queue, grid, duration.
usage: kernel_instances.py <trace dir> <pattern> [<pattern> ...]"""
import csv, glob, os, sys
root, pats = sys.argv[1], sys.argv[2:]
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
lo, hi = marks[first], marks[first + 2]
t0 = int(rows[lo]["Start_Timestamp"])
for pat in pats:
    print("==", pat)
    for r in rows[lo:hi]:
        if pat in r["Kernel_Name"]:
            print("  q%s  at %8.1f us  grid %8s x %-5s wg %4s  %8.1f us   %s" % (
                r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"],
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].replace("void ", "")[:60]))
