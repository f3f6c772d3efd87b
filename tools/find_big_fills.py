"""Where do the large hipMemsetAsync kernels of a traced run sit?  (diagnosis helper: rocprofv3 kernel-trace csv)"""
import csv, glob, os, sys
root = sys.argv[1]
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "feat_knn_pc" in r["Kernel_Name"]]
t_first_step, t_last = marks[0], marks[-1]
by_q = {}
for i, r in enumerate(rows):
    by_q.setdefault(r["Queue_Id"], []).append(i)
pos = {i: (q, j) for q, l in by_q.items() for j, i in enumerate(l)}
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "fillBuffer" in r["Kernel_Name"] and d > 60:
        q, j = pos[i]
        l = by_q[q]
        prev = rows[l[j - 1]]["Kernel_Name"][:70] if j else "-"
        nxt = rows[l[j + 1]]["Kernel_Name"][:70] if j + 1 < len(l) else "-"
        t = int(r["Start_Timestamp"])
        k = sum(1 for m in marks if m <= t)
        print("%.0f us  queue %s stream %s  after %d kNN marks (8 per step) grid %s\n     prev: %s\n     next: %s" %
              (d, q, r["Stream_Id"], k, r["Grid_Size_X"], prev, nxt))
