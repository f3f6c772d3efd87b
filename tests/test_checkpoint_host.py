"""Checkpoint files in the reference's format (models/PDGNet_v2.py:331-408) -- layout pinned by
tests/golden/checkpoint_manifest.json, generated from the reference classes under nn.DataParallel + Adam."""
import json
import os

import pytest
import torch

from pdgn_amd import fused
from pdgn_amd.trainer import PDGNTrainer

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture()
def trainer(monkeypatch):
    monkeypatch.setattr(fused, "flush_bn_counters", lambda: None)
    import pdgn_amd.trainer as T
    monkeypatch.setattr(T, "flush_bn_counters", lambda: None)
    torch.manual_seed(0)
    t = PDGNTrainer(device="cpu", distributed=False)
    for opt, mod in [(t.optG, t.G)] + list(zip(t.optD, t.D)):
        for i, p in enumerate(mod.parameters()):
            p.grad = torch.full_like(p, 1e-3 * (1 + i % 7))
        opt.step()
        opt.step()
    return t


def _check_entry(model_sd, opt_sd, man):
    assert [[k, list(v.shape)] for k, v in model_sd.items()] == man["model_keys"]
    assert len(opt_sd["param_groups"]) == man["n_param_groups"]
    g = opt_sd["param_groups"][0]
    assert g["params"] == man["group_params"]
    for k, v in man["group_hyper"].items():
        assert (list(g[k]) if isinstance(v, list) else g[k]) == v, k
    assert sorted(opt_sd["state"][0].keys()) == man["state_entry_keys"]
    assert [list(opt_sd["state"][i]["exp_avg"].shape) for i in range(len(man["state_shapes"]))] == man["state_shapes"]
    assert isinstance(opt_sd["state"][0]["step"], int) and opt_sd["state"][0]["step"] == 2


def test_saved_files_have_reference_layout(trainer, tmp_path):
    with open(os.path.join(GOLDEN, "checkpoint_manifest.json")) as f:
        man = json.load(f)
    pg, pd = trainer.save(str(tmp_path), 7, "chair")
    assert os.path.basename(pg) == "7_chair_G.pth" and os.path.basename(pd) == "7_chair_D.pth"
    g, d = torch.load(pg), torch.load(pd)
    assert list(g.keys()) == man["G_file"]["keys"] and g["G_epoch"] == 7
    assert sorted(d.keys()) == sorted(man["D_file"]["keys"]) and d["D_epoch"] == 7
    _check_entry(g["G_model"], g["G_optimizer"], man["G_file"]["G"])
    for i in (1, 2, 3, 4):
        _check_entry(d["D_model%d" % i], d["D_optimizer%d" % i], man["D_file"]["D%d" % i])


def test_roundtrip_restores_models_and_adam(trainer, tmp_path, monkeypatch):
    pg, pd = trainer.save(str(tmp_path), 3)
    torch.manual_seed(1)
    other = PDGNTrainer(device="cpu", distributed=False)
    assert other.load(pg, pd) == 3
    for a, b in [(trainer.G, other.G)] + list(zip(trainer.D, other.D)):
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb and torch.equal(va, vb), ka
    for oa, ob in [(trainer.optG, other.optG)] + list(zip(trainer.optD, other.optD)):
        sa, sb = oa.state_dict(), ob.state_dict()
        for i in sa["state"]:
            assert torch.equal(sa["state"][i]["exp_avg_sq"], sb["state"][i]["exp_avg_sq"])
            assert float(sa["state"][i]["step"]) == float(sb["state"][i]["step"]) == 2
    # the next Adam step is identical
    for opt, mod in ((trainer.optG, trainer.G), (other.optG, other.G)):
        for p in mod.parameters():
            p.grad = torch.full_like(p, 2e-3)
        opt.step()
    for pa, pb in zip(trainer.G.parameters(), other.G.parameters()):
        assert torch.equal(pa, pb)


def test_load_missing_file_raises(trainer, tmp_path):
    with pytest.raises(FileNotFoundError):
        trainer.load(str(tmp_path / "nope_G.pth"), str(tmp_path / "nope_D.pth"))
