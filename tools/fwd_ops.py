#!/usr/bin/env python3
"""Which torch ops (not the package's own kernels) does one generator forward still launch?  (launch-count hygiene)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pdgn_amd.trainer import PDGNTrainer, noise
B = 35
tr = PDGNTrainer(device="cuda"); tr.train()
z = noise(B, "cuda")
grad = len(sys.argv) > 1 and sys.argv[1] == "grad"
for _ in range(3):
    with torch.set_grad_enabled(grad):
        tr.G(z)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    with torch.set_grad_enabled(grad):
        tr.G(z)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.self_device_time_total > 0 and not e.key.startswith(("void ", "Cijk", "cl_", "wgs_", "small_mlp", "assemble", "sample_bias", "bn_softmax", "feat_knn", "csr_", "gemm_tn", "__amd", "sqnorm", "Memcpy", "Memset")):
        rows.append((e.count, e.self_device_time_total, e.key, str(e.input_shapes)[:110]))
rows.sort(key=lambda r: -r[0])
print("op, launches, self device us, shapes  (grad=%s)" % grad)
for c, t, k, s in rows[:60]:
    print("%4d  %8.1f  %-28s %s" % (c, t, k, s))
