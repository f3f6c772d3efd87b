"""CPU: host logic of the generator / discriminators / G+D step against the golden vectors of the
imported reference, with torch / C-oracle stand-ins in place of the HIP entry points."""
import json
import os

import numpy as np
import pytest
import torch

from hashweights import fill_module, hash_tensor
from torch_standins import EdgeGatherSumTorch, linear_cl_torch, softmax_slots_permute_torch, bn_softmax_slots_permute_torch, bilateral_weighting_torch, bn_act_maxpool_torch, bn_act_torch, feature_knn_torch, patch_losses

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def patched(monkeypatch):
    from pdgn_amd import deconv
    monkeypatch.setattr(deconv, "EdgeGatherSum", EdgeGatherSumTorch)
    monkeypatch.setattr(deconv, "bn_act", bn_act_torch)
    monkeypatch.setattr(deconv, "bn_act_maxpool", bn_act_maxpool_torch)
    monkeypatch.setattr(deconv, "linear_cl", linear_cl_torch)
    monkeypatch.setattr(deconv, "flush_bn_counters", lambda: None)
    monkeypatch.setattr(deconv, "softmax_slots_permute", softmax_slots_permute_torch)
    monkeypatch.setattr(deconv, "bn_softmax_slots_permute", bn_softmax_slots_permute_torch)
    monkeypatch.setattr(deconv, "bilateral_weighting", bilateral_weighting_torch)
    monkeypatch.setattr(deconv, "feature_knn", feature_knn_torch)
    return deconv


def test_state_dict_keys_match_reference(patched):
    from pdgn_amd.generator import PointDiscriminator, PointGenerator
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        man = json.load(f)
    assert {k: list(v.shape) for k, v in PointGenerator().state_dict().items()} == man["G"]
    for i in (1, 2, 3, 4):
        assert {k: list(v.shape) for k, v in PointDiscriminator(i).state_dict().items()} == man["D%d" % i]
    assert sum(p.numel() for p in PointGenerator().parameters()) == 12711372   # BASELINE.md probe


def test_generator_and_discriminators_forward(golden, patched):
    from pdgn_amd.generator import PointDiscriminator, PointGenerator
    g = golden("generator_b6.npz")
    G = fill_module(PointGenerator(), salt=1).train()
    with torch.no_grad():
        outs = G(torch.from_numpy(g["z"]),
                 idx=[torch.from_numpy(g["idx%d" % i].astype(np.int32)) for i in (1, 2, 3, 4)])
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), g["p%d" % (i + 1)], rtol=1e-3, atol=1e-4)
    for i in (1, 2, 3, 4):
        D = fill_module(PointDiscriminator(i), salt=9 + i).train()
        with torch.no_grad():
            np.testing.assert_allclose(D(torch.from_numpy(g["p%d" % i])).numpy(), g["d%d" % i],
                                       rtol=1e-4, atol=1e-5)


def test_load_reference_checkpoint_prefix(patched):
    from pdgn_amd.generator import PointDiscriminator, load_reference_state_dict
    src = fill_module(PointDiscriminator(2), salt=4)
    dst = PointDiscriminator(2)
    load_reference_state_dict(dst, {"module." + k: v for k, v in src.state_dict().items()})
    for a, b in zip(src.state_dict().values(), dst.state_dict().values()):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,rtol", [(4, 2e-3), (2, 0.1)])
def test_one_step_matches_composed_reference(golden, patched, monkeypatch, B, rtol):
    """The six logged losses and an Adam-updated weight slice of one iteration equal the
    composed-reference fixture.  B=2 is BASELINE.json configs[0]'s batch size: BatchNorm1d over 2
    samples amplifies fp32 rounding ~100x per stage (the fp32 reference itself is 5e-3 away from
    its own fp64 evaluation there), hence the loose tolerance; B=4 is the tight check."""
    from oracle import pdgnet_ref
    from pdgn_amd.trainer import PDGNTrainer
    g = golden("step_b%d.npz" % B)
    tr = PDGNTrainer(device="cpu", distributed=False)
    fill_module(tr.G, salt=1)
    for i, d in enumerate(tr.D):
        fill_module(d, salt=10 + i)

    patch_losses(monkeypatch)                     # C-oracle kNN + torch stats/Chamfer for the HIP ops
    tr.train()
    reals = [hash_tensor("real%d" % i, (B, 3, n), 0.8) for i, n in enumerate((256, 512, 1024, 2048))]
    out = tr.step(reals, hash_tensor("step_z1", (B, 128), 0.2), hash_tensor("step_z2", (B, 128), 0.2))
    for key in ("d_loss1", "d_loss2", "d_loss3", "d_loss4", "g_loss", "similar_loss"):
        np.testing.assert_allclose(out[key].item(), g[key], rtol=rtol, err_msg=key)
    np.testing.assert_allclose(tr.G.fc1[0].weight.detach()[:4, :8].numpy(), g["g_fc1_w_after"],
                               rtol=1e-3, atol=2e-4 if B == 2 else 2e-6)


def test_one_step_b8_backward_against_the_references_gradients(golden, patched, monkeypatch):
    """Host logic of the BACKWARD (weight re-association, constant-channel split, gather-sum adjoint, zero bias gradients,
    frozen discriminators) against the reference's own gradients: tests/golden/step_b8_graphs.npz with the reference's
    eight kNN graphs forced in, so that what is left is arithmetic (step_checks.check_step_gradients)."""
    from pdgn_amd.trainer import PDGNTrainer
    from step_checks import check_step_gradients
    g = golden("step_b8_graphs.npz")
    graphs = [torch.from_numpy(g["graph%d" % i].astype(np.int32)) for i in range(8)]
    calls = []

    def forced(x, k):
        calls.append(tuple(x.shape))
        return graphs[len(calls) - 1]
    monkeypatch.setattr(patched, "feature_knn", forced)
    tr = PDGNTrainer(device="cpu", distributed=False)
    fill_module(tr.G, salt=1)
    for i, d in enumerate(tr.D):
        fill_module(d, salt=10 + i)
    patch_losses(monkeypatch)
    tr.train()
    B = 8
    reals = [hash_tensor("real%d" % i, (B, 3, n), 0.8) for i, n in enumerate((256, 512, 1024, 2048))]
    out = tr.step(reals, hash_tensor("step_z1", (B, 128), 0.2), hash_tensor("step_z2", (B, 128), 0.2))
    assert len(calls) == 8
    for key in ("d_loss1", "d_loss2", "d_loss3", "d_loss4", "g_loss", "similar_loss"):
        np.testing.assert_allclose(out[key].item(), float(g[key]), rtol=2e-3, err_msg=key)
    check_step_gradients(tr, g)
