"""Measured error of the deconvolution blocks' backward against the imported reference's fixtures (tests/golden/deconv_*.npz):
for grad_x, grad_pc and every grad.<param>: max |ours - reference| / max |reference|, for
  host32 : the re-associated host logic with torch stand-ins for the HIP entry points, fp32, CPU
  host64 : the same in fp64 (the arithmetic-free answer: what is left is the REFERENCE's own fp32 rounding)
  x3 / x3_16 / fp32 : the HIP kernels (GPU only), every dense layer on the own kernels (fused._OWN_MIN_ROWS = 1)
VERDICT r4 #5: the tests held these at rtol 1e-3 with no measured reason.
    python tools/backward_error.py            (CPU arms; adds the GPU arms when a GPU is present)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from hashweights import fill_module

NAMES = ["plain_k4", "bilateral_k4", "plain_k10", "bilateral_k10"]


def rel(a, b):
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def run(name, arm):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "deconv_%s.npz" % name)))
    from pdgn_amd import deconv, fused
    bilateral = name.startswith("bilateral")
    gpu = arm in ("x3", "x3_16", "fp32")
    saved = {}
    if not gpu:
        import torch_standins as ts
        for k, v in (("EdgeGatherSum", ts.EdgeGatherSumTorch), ("bn_act", ts.bn_act_torch), ("bn_act_maxpool", ts.bn_act_maxpool_torch),
                     ("linear_cl", ts.linear_cl_torch), ("flush_bn_counters", lambda: None),
                     ("softmax_slots_permute", ts.softmax_slots_permute_torch), ("bn_softmax_slots_permute", ts.bn_softmax_slots_permute_torch),
                     ("bilateral_weighting", ts.bilateral_weighting_torch)):
            saved[k] = getattr(deconv, k)
            setattr(deconv, k, v)
    else:
        from pdgn_amd import _lib
        _lib.set_gemm_mode(arm)
        saved_rows, fused._OWN_MIN_ROWS = fused._OWN_MIN_ROWS, 1
    try:
        dt = torch.float64 if arm == "host64" else torch.float32
        dev = "cuda" if gpu else "cpu"
        mod = fill_module(deconv.PointDeconv(int(g["F"]), int(g["Fout"]), int(g["k"]), bilateral=bilateral), salt=3).to(dev).to(dt)
        x = torch.from_numpy(g["x"]).to(dev).to(dt).requires_grad_(True)
        pc = torch.from_numpy(g["pc"]).to(dev).to(dt).requires_grad_(True) if bilateral else None
        idx = torch.from_numpy(g["idx"].astype(np.int32)).to(dev)
        mod.train()
        y = mod(x, pc, idx=idx)
        y.backward(torch.from_numpy(g["gout"]).to(dev).to(dt))
        out = {"y": rel(y.detach().cpu().numpy(), g["y_train"]), "grad_x": rel(x.grad.cpu().numpy(), g["grad_x"])}
        if bilateral:
            out["grad_pc"] = rel(pc.grad.cpu().numpy(), g["grad_pc"])
        # parameter gradients: relative to the parameter's own largest gradient; the biases in front of a training-mode
        # BatchNorm have an identically zero gradient (the reference holds 1e-9 of rounding residue there, this code exact
        # zeros): those are reported as an absolute residue relative to the block's largest weight gradient
        gmax = max(np.abs(g["grad." + n]).max() for n, _ in mod.named_parameters())
        worst, resid = ("", 0.0), 0.0
        for n, p in mod.named_parameters():
            ref = g["grad." + n]
            if np.abs(ref).max() < 1e-6 * gmax:
                resid = max(resid, float(np.abs(p.grad.cpu().numpy().astype(np.float64) - ref).max() / gmax))
                continue
            e = rel(p.grad.cpu().numpy(), ref)
            if e > worst[1]:
                worst = (n, e)
        out["worst grad.<param>"] = worst[1]
        out["(which)"] = worst[0]
        out["zero-gradient biases, residue / largest weight gradient"] = resid
        return out
    finally:
        if not gpu:
            for k, v in saved.items():
                setattr(deconv, k, v)
        else:
            fused._OWN_MIN_ROWS = saved_rows
            _lib.set_gemm_mode("x3")


arms = ["host32", "host64"] + (["x3", "x3_16", "fp32"] if torch.cuda.is_available() else [])
print("max |ours - reference fixture| / max |reference|   (fixtures: the imported reference's fp32 blocks on CPU)")
for name in NAMES:
    for arm in arms:
        o = run(name, arm)
        print("%-14s %-7s " % (name, arm) + "  ".join("%s %.2e" % (k, v) if not isinstance(v, str) else "%s %s" % (k, v) for k, v in o.items()))
