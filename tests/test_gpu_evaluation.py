"""All-pairs evaluation on the pair-list kernels (through the C ABI) against the reference's
distChamfer fixture and the C oracle of approxmatch/matchcost."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def test_pairwise_cd_golden_and_emd_oracle(golden):
    from oracle import cref
    from pdgn_amd import evaluation as ev
    g = golden("eval_metrics.npz")
    smp = torch.from_numpy(g["smp"]).to(_dev())
    ref = torch.from_numpy(g["ref"]).to(_dev())
    cd, emd = ev.pairwise_emd_cd(smp, ref)
    np.testing.assert_allclose(cd.cpu().numpy(), g["all_cd"], rtol=1e-4, atol=1e-6)   # Gram form, fp32
    S, R = g["smp"].shape[0], g["ref"].shape[0]
    want = np.zeros((S, R), np.float32)
    for i in range(S):
        want[i] = cref.emd_approx(np.repeat(g["smp"][i:i + 1], R, axis=0), g["ref"])
    np.testing.assert_allclose(emd.cpu().numpy(), want, rtol=1e-4, atol=1e-6)


def test_indexed_equals_expanded_bitwise():
    """pair lists must give exactly what the reference's expand-and-call gives"""
    from pdgn_amd import evaluation as ev
    from pdgn_amd.losses import chamfer_min
    from pdgn_amd.structural_losses import emd_cost
    torch.manual_seed(3)
    smp = torch.randn(7, 256, 3, device=_dev())
    ref = torch.randn(5, 256, 3, device=_dev())
    cd, emd = ev.pairwise_emd_cd(smp, ref)
    for i in range(7):
        a = smp[i:i + 1].expand(5, -1, -1).contiguous()
        minx, miny = chamfer_min(a, ref)
        assert torch.equal(cd[i], miny.mean(1) + minx.mean(1))
        assert torch.equal(emd[i], emd_cost(a, ref) / 256.0)


def test_pairwise_chunking(monkeypatch):
    from pdgn_amd import evaluation as ev
    torch.manual_seed(4)
    smp = torch.randn(9, 64, 3, device=_dev())
    ref = torch.randn(11, 64, 3, device=_dev())
    cd0, emd0 = ev.pairwise_emd_cd(smp, ref)
    monkeypatch.setattr(ev, "_MAX_PAIRS", 13)
    cd1, emd1 = ev.pairwise_emd_cd(smp, ref)
    assert torch.equal(cd0, cd1) and torch.equal(emd0, emd1)


def test_compute_all_metrics_keys_and_sanity():
    from pdgn_amd import evaluation as ev
    torch.manual_seed(5)
    ref = torch.randn(12, 128, 3, device=_dev())
    res = ev.compute_all_metrics(ref.clone(), ref)
    want = {"lgan_mmd-CD", "lgan_cov-CD", "lgan_mmd_smp-CD", "lgan_mmd-EMD", "lgan_cov-EMD", "lgan_mmd_smp-EMD",
            "1-NN-CD-acc_t", "1-NN-CD-acc_f", "1-NN-CD-acc", "1-NN-EMD-acc_t", "1-NN-EMD-acc_f", "1-NN-EMD-acc"}
    assert set(res) == want
    # identical sets: every reference cloud is matched by itself -> full coverage, ~zero MMD
    assert float(res["lgan_cov-CD"]) == 1.0 and float(res["lgan_cov-EMD"]) == 1.0
    assert float(res["lgan_mmd-CD"]) < 1e-5
    r2 = ev.emd_cd(ref, ref)
    assert float(r2["MMD-CD"]) < 1e-5


def test_emd_cd_paired_matches_oracle():
    from oracle import cref
    from pdgn_amd import evaluation as ev
    torch.manual_seed(6)
    a = torch.randn(4, 128, 3)
    b = torch.randn(4, 128, 3)
    r = ev.emd_cd(a.to(_dev()), b.to(_dev()), reduced=False)
    d1, _, d2, _ = cref.nndistance(a.numpy(), b.numpy())
    np.testing.assert_allclose(r["MMD-CD"].cpu().numpy(), d1.mean(1) + d2.mean(1), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(r["MMD-EMD"].cpu().numpy(), cref.emd_approx(a.numpy(), b.numpy()), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("res", [8, 28])
def test_jsd_matches_reference(golden, res):
    """occupancy counters (one brute-force 1-NN launch over the sphere-clipped grid) and JSD against the imported
    reference's sklearn / scipy implementation (tests/golden/data_jsd.npz)."""
    from pdgn_amd import evaluation as ev
    g = golden("data_jsd.npz")
    smp = torch.from_numpy(g["jsd_smp"]).to(_dev())
    ref = torch.from_numpy(g["jsd_ref"]).to(_dev())
    ent, counters = ev.entropy_of_occupancy_grid(smp, res, True)
    np.testing.assert_array_equal(counters.cpu().numpy(), g["counters_%d" % res])
    np.testing.assert_allclose(float(ent), float(g["ent_%d" % res]), rtol=1e-9)
    np.testing.assert_allclose(float(ev.jsd_between_point_cloud_sets(smp, ref, resolution=res)), float(g["jsd_%d" % res]),
                               rtol=1e-9, atol=1e-12)
    assert float(ev.jsd_between_point_cloud_sets(smp, smp, resolution=res)) < 1e-12


def test_generate_and_evaluate_test_phase():
    """the reference's test() flow end to end on a small generator: shapes, keys, finite values, determinism of the
    seeded noise, and that a set evaluated against itself has full coverage"""
    from pdgn_amd import evaluation as ev
    from pdgn_amd.data import normalize_clouds
    from pdgn_amd.generator import PointGenerator
    torch.manual_seed(2)
    G = PointGenerator(base_points=16).to(_dev()).eval()
    ref = normalize_clouds(torch.randn(10, 256, 3, device=_dev()) * 0.3, "shape_bbox")[0] * 0.45
    rng = torch.Generator().manual_seed(5)
    gen, res = ev.generate_and_evaluate(G, ref, batch_size=4, normalize="shape_bbox", rng=rng)
    assert gen.shape == (10, 256, 3)
    assert abs(float(gen.abs().max()) - 1.0) < 1e-5                         # shape_bbox: longest side spans [-1, 1]
    assert {"lgan_mmd-CD", "lgan_cov-EMD", "1-NN-CD-acc", "1-NN-EMD-acc", "jsd"} <= set(res)
    assert all(torch.isfinite(torch.as_tensor(v)).all() for v in res.values())
    gen2, _ = ev.generate_and_evaluate(G, ref, batch_size=4, normalize="shape_bbox", rng=torch.Generator().manual_seed(5),
                                       with_jsd=False)
    assert torch.equal(gen, gen2)


def test_pairwise_sharded_over_a_one_rank_rccl_group_equals_unsharded():
    """The rank-sharded all-pairs path (SURVEY.md 8-e eval row) through RCCL's all-gather at world size 1."""
    import os
    import socket
    import torch.distributed as dist
    from pdgn_amd import evaluation
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    g = torch.Generator().manual_seed(5)
    a = torch.rand(5, 256, 3, generator=g).cuda() * 2 - 1
    b = torch.rand(7, 256, 3, generator=g).cuda() * 2 - 1
    cd0, emd0 = evaluation.pairwise_emd_cd(a, b)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cd1, emd1 = evaluation.pairwise_emd_cd(a, b, shard_over_ranks=True)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert torch.equal(cd0, cd1) and torch.equal(emd0, emd1)
