#!/usr/bin/env python3
"""Where the torch glue launches of one G+D step come from: aten ops that launch fill / copy / add / reduce kernels, grouped by
the innermost pdgn_amd source line on their Python stack (ops issued by the autograd engine itself -- gradient accumulation,
materialised zero gradients -- have no Python frame and are listed under their autograd node)."""
import os, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
B = int(os.environ.get("B", "35"))
tr = PDGNTrainer(device="cuda", distributed=False); tr.train()
reals = synthetic_batch(B, "cuda")
for _ in range(3):
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(reals, noise(B, "cuda"), noise(B, "cuda"))
    torch.cuda.synchronize()
ev = prof.events()
WANT = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::mul_", "aten::cat",
        "aten::leaky_relu", "aten::leaky_relu_backward", "aten::neg", "aten::mean", "aten::mse_loss", "aten::mse_loss_backward",
        "aten::div", "aten::sub", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zeros_like", "aten::constant_pad_nd",
        "aten::mm", "aten::addmm", "aten::linear", "aten::matmul")
LEAF = ("aten::fill_", "aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::mul_", "aten::cat", "aten::leaky_relu",
        "aten::leaky_relu_backward", "aten::neg", "aten::mean", "aten::mse_loss", "aten::mse_loss_backward", "aten::div", "aten::sub",
        "aten::mm", "aten::addmm")
cnt, dev, shapes = Counter(), Counter(), {}
for e in ev:
    if e.name not in LEAF:
        continue
    kt = sum(k.duration for k in e.kernels) if e.kernels else 0.0
    if not e.kernels:
        continue
    where = None
    for fr in e.stack or []:
        if "pdgn_amd" in fr and "site-packages" not in fr:
            where = fr.split("pdgn_amd/")[-1].strip()
            break
    if where is None:
        p = e.cpu_parent
        chain = []
        while p is not None and len(chain) < 3:
            chain.append(p.name[:48])
            p = p.cpu_parent
        where = "engine: " + " < ".join(chain) if chain else "engine"
    key = (e.name, where)
    cnt[key] += 1
    dev[key] += kt
    shapes.setdefault(key, Counter())[str(e.input_shapes)[:60]] += 1
print("aten leaf ops with kernels: %d launches-ish, %.0f us of kernel time" % (sum(cnt.values()), sum(dev.values())))
for key, c in sorted(cnt.items(), key=lambda kv: -dev[kv[0]])[:70]:
    print("%4d x %8.1f us  %-28s %s   %s" % (c, dev[key], key[0], key[1][:90], shapes[key].most_common(1)[0][0]))
