"""Which logical streams of the step share a hardware queue?  (rocprofv3 --kernel-trace csv -> marker kernels per queue.)

HIP multiplexes streams onto GPU_MAX_HW_QUEUES (4) hardware queues; streams on one queue serialise.  The markers: the
gather-sum adjoint (generator backward: the default stream), feat_knn (kNN side stream),
chamfer (local-pair stream), the discriminators' max-pool kernels by grid size (D1..D4 streams), RCCL kernels."""
import csv, glob, os, sys
from collections import Counter, defaultdict

root = sys.argv[1]
path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(path)))
marks = sorted(int(r["Start_Timestamp"]) for r in rows if "feat_knn_pc" in r["Kernel_Name"])
t0, t1 = marks[-16], marks[-8]
tags = (("wgs_bwd_csr", "G backward (default stream)"), ("feat_knn_pc", "feature kNN stream"),
        ("chamfer_gram_grad", "local-pair stream"), ("ncclDevKernel", "RCCL"), ("rccl", "RCCL"))
dlevel = {"8960": "D1", "17920": "D2 or D3", "35840": "D4"}      # B x width of the layer in front of the max-pool: 256 / 512 / 512 / 1024
q = defaultdict(Counter)
n = Counter()
for r in rows:
    if not t0 <= int(r["Start_Timestamp"]) < t1:
        continue
    k, name = (r["Queue_Id"], r["Stream_Id"]), r["Kernel_Name"]
    n[k] += 1
    for sub, tag in tags:
        if sub in name:
            q[k][tag] += 1
    if "cl_max_finalize" in name:
        q[k][dlevel.get(r["Grid_Size_X"], "D?" + r["Grid_Size_X"]) + " stream"] += 1
print("one step; (hardware queue, HIP stream id): launches, markers")
for k in sorted(n):
    print("  queue %s stream %s: %4d launches  %s" % (k[0], k[1], n[k], dict(q[k])))
