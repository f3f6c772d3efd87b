"""Every library / torch-native kernel of one eager iteration (at::native::*, Cijk_* = hipBLASLt / rocBLAS, fills and
copies) with the torch operator that launched it and the pdgn_amd line that called that operator.
    python tools/glue_owners.py [B=35]"""
import os
import sys
from collections import Counter, defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 35
dev = torch.device("cuda", 0)
torch.manual_seed(9999)
tr = PDGNTrainer(device=dev, distributed=False)
tr.train()
reals = synthetic_batch(B, dev)
g = torch.Generator().manual_seed(1234)
zs = [(noise(B, dev, g), noise(B, dev, g)) for _ in range(4)]
for i in range(3):
    tr.step(reals, *zs[i])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(reals, *zs[3])
    torch.cuda.synchronize()
own = ("gemm_x3", "gemm_nt", "gemm_tn", "wgs_", "cl_", "knn", "thin_", "skinny", "small_mlp", "bilateral", "bn_softmax", "softmax_perm",
       "chamfer", "local_stats", "csr_", "cf_", "group_colsum", "pointmax", "assemble", "split_bf16", "sample_bias", "mse_", "scaled_sum",
       "feat_knn", "replay_marker", "nn3", "grouping", "interp", "emd_", "nndist", "spin", "colsum", "deconv_")
cnt, dur, who = Counter(), Counter(), defaultdict(Counter)
for e in prof.events():
    ks = getattr(e, "kernels", None)
    if not ks or e.device_type != torch.autograd.DeviceType.CPU:
        continue
    for k in ks:
        name = k.name.replace("void ", "")
        if any(t in name for t in own):
            continue
        short = name.split("<")[0].split("(")[0][:60] if not name.startswith("Cijk") else name[:70]
        frames = [f.split("/")[-1] for f in (e.stack or []) if "pdgn_amd" in f]
        # the owner: the Python line when the profiler has it, else the nearest enclosing non-aten event (an autograd Function's
        # forward, or "autograd::engine::evaluate_function: <Node>" for a backward)
        anc, p = None, e.cpu_parent
        while p is not None:
            if not (p.name.startswith("aten::") or p.name.startswith("hip")):
                anc = p.name.replace("autograd::engine::evaluate_function: ", "bwd ")
                break
            p = p.cpu_parent
        key = (short, e.name)
        cnt[key] += 1
        dur[key] += k.duration
        who[key][frames[0][:70] if frames else (anc or "(top level: trainer / generator Python)")] += 1
tot = sum(cnt.values())
print("B %d: %d library / torch-native launches in one iteration, %.0f us of kernel time" % (B, tot, sum(dur.values())))
for key, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print("%3d x %7.1f us  %-62s %-22s %s" % (n, dur[key], key[0], key[1], "; ".join("%s x%d" % (w, c) for w, c in who[key].most_common(6))))
