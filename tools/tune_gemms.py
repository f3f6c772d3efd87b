#!/usr/bin/env python3
"""Offline search over all rocBLAS / hipBLASLt solutions (torch TunableOp) for the LARGE GEMM shapes of the G+D step,
in situ (same operand layouts as the step issues them).  Writes pdgn_amd/tunableop_gfx950.csv-style output to
gpurun_out/tunableop_gfx950.csv; commit it as pdgn_amd/tunableop_gfx950.csv.  Run on an MI355X:
    PDGN_BLAS_TUNABLE_SEARCH=1 python tools/tune_gemms.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("PDGN_BLAS_TUNABLE_SEARCH") == "1", "set PDGN_BLAS_TUNABLE_SEARCH=1"
os.environ["PDGN_OVERLAP"] = "0"                     # sequential schedule: every GEMM is timed on an otherwise idle device
os.makedirs("gpurun_out", exist_ok=True)
out = os.path.abspath("gpurun_out/tunableop_gfx950.csv")
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = out
import torch
import torch.cuda.tunable as tn
tn.set_filename(out, False)
tn.set_max_tuning_duration(int(os.environ.get("PDGN_TUNE_MS", "15")))
tn.set_max_tuning_iterations(20)
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = 35
t0 = time.time()
for base in (128, 256):                                  # the reference's 256->2048 configuration and the 512->4096 one (C4)
    res_pts = tuple((2 * base) << i for i in range(4))
    tr = PDGNTrainer(device="cuda", base_points=base); tr.train()
    reals = synthetic_batch(B, "cuda", n_points=res_pts[3], resolutions=res_pts)
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))   # first iteration: fused._library_gemm searches every shape it meets
    torch.cuda.synchronize()
    del tr, reals
    torch.cuda.empty_cache()
print("search took %.0f s" % (time.time() - t0))
res = tn.get_results()
print("%d tuned entries" % len(res))
with open(out, "w") as f:
    for v in tn.get_validators():
        f.write("Validator,%s,%s\n" % (v[0], v[1]))
    for r in res:
        f.write(",".join(str(x) for x in r) + "\n")
print(open(out).read()[:3000])
