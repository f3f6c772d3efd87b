#!/usr/bin/env python3
"""Op-level attribution of one G+D step (torch.profiler, device time, grouped by op + input shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 35
tr = PDGNTrainer(device="cuda"); tr.train()
reals = synthetic_batch(B, "cuda")
for _ in range(3):
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
print(ka.table(sort_by="device_time_total", row_limit=70, max_name_column_width=40, max_shapes_column_width=70))
print("==== glue ops (self device time, by shape) ====")
glue = [e for e in ka if e.key.startswith("aten::") and e.self_device_time_total > 0 and not any(
    t in e.key for t in ("mm", "linear", "matmul"))]
glue.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in glue)
print("total glue self device time: %.3f ms" % (tot / 1e3))
for e in glue[:80]:
    print("%-28s %4d calls %9.1f us  %s" % (e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:110]))
