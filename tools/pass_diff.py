"""Default-stream kernels of generator pass #1 (no grad) vs pass #2 (grad) from a rocprofv3 kernel trace: what does
building the autograd graph add to the forward?  Pass boundaries: the D1 update's first kernel does not exist on the
default stream, so the passes are cut at the small_mlp_fwd kernel of fc1 (first kernel of a generator pass)."""
import csv, glob, os, sys
from collections import Counter
root = sys.argv[1]
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = [r for r in csv.DictReader(open(path)) if r["Stream_Id"] == "0"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# generator passes start with the fc1 small_mlp_fwd launch whose grid is the largest of the pass (128*32*... channels)
starts = [i for i, r in enumerate(rows) if "small_mlp_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 4096 * 64]
print("pass starts found:", len(starts))
def seg(i0, i1):
    c, t = Counter(), Counter()
    for r in rows[i0:i1]:
        n = r["Kernel_Name"].split("(")[0][:60]
        c[n] += 1
        t[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return c, t
# last complete step: passes -4 (z1), -3 (z2) of the last two steps
a0, a1, a2 = starts[-4], starts[-3], starts[-2]
# pass 2 ends where the backward begins: first bilateral_bwd / gemm_tn kernel after a1
end2 = next(i for i in range(a1, a2) if "gemm_tn" in rows[i]["Kernel_Name"] or "_bwd" in rows[i]["Kernel_Name"])
c1, t1 = seg(a0, a1)
c2, t2 = seg(a1, end2)
print("pass1: %d kernels %.2f ms busy | pass2 (to first backward kernel): %d kernels %.2f ms busy" % (sum(c1.values()), sum(t1.values()) / 1e3, sum(c2.values()), sum(t2.values()) / 1e3))
for n in sorted(set(c1) | set(c2), key=lambda n: -(abs(t2[n] - t1[n]))):
    if c1[n] != c2[n] or abs(t2[n] - t1[n]) > 20:
        print("  %-62s pass1 %3d x %8.1f us   pass2 %3d x %8.1f us" % (n, c1[n], t1[n], c2[n], t2[n]))
if len(sys.argv) > 2:
    print("pass 1, all kernels (count, total us):")
    for n, k in c1.most_common():
        print("  %3d x %8.1f us  %s" % (k, t1[n], n))
    b0 = end2
    b1 = next(i for i in range(b0, len(rows)) if "multi_tensor" in rows[i]["Kernel_Name"])
    cb, tb = seg(b0, b1)
    print("backward on the default stream: %d kernels %.2f ms busy" % (sum(cb.values()), sum(tb.values()) / 1e3))
    for n, k in cb.most_common():
        print("  %3d x %8.1f us  %s" % (k, tb[n], n))
