"""CPU: host logic of PointDeconv (weight re-association + layouts) against the golden vectors
of the imported reference, with torch stand-ins in place of the two HIP entry points."""
import numpy as np
import pytest
import torch

from hashweights import fill_module
from step_checks import assert_block_gradients
from torch_standins import EdgeGatherSumTorch, linear_cl_torch, softmax_slots_permute_torch, bn_softmax_slots_permute_torch, bilateral_weighting_torch, bn_act_maxpool_torch, bn_act_torch


@pytest.fixture()
def deconv(monkeypatch):
    from pdgn_amd import deconv as m
    monkeypatch.setattr(m, "EdgeGatherSum", EdgeGatherSumTorch)
    monkeypatch.setattr(m, "bn_act", bn_act_torch)
    monkeypatch.setattr(m, "bn_act_maxpool", bn_act_maxpool_torch)
    monkeypatch.setattr(m, "linear_cl", linear_cl_torch)
    monkeypatch.setattr(m, "flush_bn_counters", lambda: None)
    monkeypatch.setattr(m, "softmax_slots_permute", softmax_slots_permute_torch)
    monkeypatch.setattr(m, "bn_softmax_slots_permute", bn_softmax_slots_permute_torch)
    monkeypatch.setattr(m, "bilateral_weighting", bilateral_weighting_torch)
    return m


@pytest.mark.parametrize("name", ["plain_k4", "bilateral_k4", "plain_k10", "bilateral_k10"])
def test_pointdeconv_matches_reference(golden, deconv, name):
    g = golden("deconv_%s.npz" % name)
    bilateral = name.startswith("bilateral")
    mod = deconv.PointDeconv(int(g["F"]), int(g["Fout"]), int(g["k"]), bilateral=bilateral)
    fill_module(mod, salt=3)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    pc = torch.from_numpy(g["pc"]).requires_grad_(True) if bilateral else None
    idx = torch.from_numpy(g["idx"]).to(torch.int32)
    mod.train()
    y = mod(x, pc, idx=idx)
    np.testing.assert_allclose(y.detach().numpy(), g["y_train"], rtol=1e-4, atol=2e-5)
    y.backward(torch.from_numpy(g["gout"]))
    assert_block_gradients(mod, g, x, pc)                      # 5e-6 of each tensor's scale: 2x the worst measured (step_checks.py)
    for n, b in mod.named_buffers():
        if "num_batches" in n:
            assert int(b) == 1, n
        else:
            np.testing.assert_allclose(b.numpy(), g["stat." + n], rtol=1e-4, atol=1e-5, err_msg=n)
    mod.eval()
    with torch.no_grad():
        np.testing.assert_allclose(mod(x, pc, idx=idx).numpy(), g["y_eval"], rtol=1e-4, atol=2e-5)


def test_linear_cl_refuses_a_bare_gradient_placeholder():
    """BNActMaxPool's closed tail hands the dense layer a zero-stride placeholder and the real gradient through a side
    table; a placeholder that arrives WITHOUT its entry (consumed elsewhere, another node in between) must raise, not be used."""
    from pdgn_amd import fused
    x = torch.randn(6, 4, requires_grad=True)
    w = torch.randn(3, 4, requires_grad=True)
    y = fused.LinearCL.apply(x, w, None, None, False, None)
    tok = torch.full((1,), float("nan")).expand(6, 3)
    with pytest.raises(RuntimeError, match="placeholder"):
        y.backward(tok)
    # ... and with its entry the carried gradients come out untouched
    fused.clear_zero_colsum()
    y = fused.LinearCL.apply(x, w, None, None, False, None)
    dh, dw = torch.ones(6, 4), torch.ones(3, 4)
    y.backward(fused._placeholder_with_input_grad(6, 3, dh, dw, x.device))
    assert torch.equal(x.grad, dh) and torch.equal(w.grad, dw) and not fused._INPUT_GRADS
