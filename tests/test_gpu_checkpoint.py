"""Checkpoint round trip on the device (capturable/fused Adam keeps `step` on the GPU)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gpu_checkpoint_roundtrip_and_resume(tmp_path):
    from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    t = PDGNTrainer(device=dev, distributed=False)
    t.train()
    reals = synthetic_batch(4, dev)
    g = torch.Generator().manual_seed(1)
    t.step(reals, noise(4, dev, g), noise(4, dev, g))
    pg, pd = t.save(str(tmp_path), 1, "chair")
    ck = torch.load(pg)
    assert all(k.startswith("module.") for k in ck["G_model"])
    assert ck["G_optimizer"]["state"][0]["step"] == 1 and not ck["G_optimizer"]["state"][0]["exp_avg"].is_cuda
    torch.manual_seed(5)
    u = PDGNTrainer(device=dev, distributed=False)
    u.train()
    assert u.load(pg, pd) == 1
    for (ka, va), (kb, vb) in zip(t.G.state_dict().items(), u.G.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    z1, z2 = noise(4, dev, g), noise(4, dev, g)
    la = [float(x) for x in t.step(reals, z1, z2).values()]
    lb = [float(x) for x in u.step(reals, z1, z2).values()]
    # same weights, Adam moments and inputs; only atomics' summation order differs between the two runs
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (la, lb)
    st = u.optG.state_dict()["state"][0]["step"]
    assert float(st) == 2.0
