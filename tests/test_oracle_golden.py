"""CPU: the oracle (oracle/) against the golden vectors produced by the imported reference
(tests/golden/gen_golden.py) and against known-answer properties.  No GPU, no pdgn_amd."""
import json
import os

import numpy as np
import pytest
import torch

from hashweights import fill_module, hash_tensor, lattice_points
from oracle import cref, pdgnet_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------ pointops
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_knnquery_matches_reference_naive(golden, tag):
    g = golden("pointops_knn.npz")
    idx, dist2 = cref.knnquery(int(g[tag + "_k"]), g[tag + "_xyz"], g[tag + "_new_xyz"])
    np.testing.assert_array_equal(idx, g[tag + "_idx"])
    assert (np.diff(dist2, axis=2) >= 0).all()


def test_knnquery_ties_and_short_sets():
    # duplicate points: strict '<' keeps the lower index first (knnquery_cuda_kernel.cu:33)
    xyz = np.zeros((1, 6, 3), np.float32)
    xyz[0, 3:] = 1.0
    idx, d2 = cref.knnquery(4, xyz, xyz[:, :1])
    np.testing.assert_array_equal(idx[0, 0], [0, 1, 2, 3])
    np.testing.assert_array_equal(d2[0, 0], [0, 0, 0, 3])
    # n < nsample: tail stays idx 0 / dist +inf (:23-26)
    idx, d2 = cref.knnquery(5, xyz[:, :2], xyz[:, 3:4])
    np.testing.assert_array_equal(idx[0, 0], [0, 1, 0, 0, 0])
    assert np.isinf(d2[0, 0, 2:]).all()


def test_grouping_fwd_bwd_vs_torch_gather():
    b, c, n, m, ns = 2, 5, 33, 7, 6
    pts = hash_tensor("grp_p", (b, c, n))
    idx = torch.from_numpy((np.abs(lattice_points("grp_i", (b, m, ns)) * 1000).astype(np.int64)) % n)
    ref = torch.gather(pts, 2, idx.view(b, 1, m * ns).expand(b, c, m * ns)).view(b, c, m, ns)
    out = cref.grouping_forward(pts.numpy(), idx.numpy())
    np.testing.assert_array_equal(out, ref.numpy())
    g = hash_tensor("grp_g", (b, c, m, ns))
    pts2 = pts.clone().requires_grad_(True)
    torch.gather(pts2, 2, idx.view(b, 1, m * ns).expand(b, c, m * ns)).view(b, c, m, ns).backward(g)
    np.testing.assert_allclose(cref.grouping_backward(g.numpy(), idx.numpy(), n), pts2.grad.numpy(),
                               rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_nearestneighbor_matches_reference_naive(golden, tag):
    """3-NN pinned to the imported reference: KNNQueryNaive.forward(None, 3, known, unknown) on lattice inputs
    (lib/pointops/functions/pointops.py:368-405 <-> :61-83); gen_golden.py::gen_nn3."""
    g = golden("pointops_nn3.npz")
    d2, idx = cref.nearestneighbor(g[tag + "_unknown"], g[tag + "_known"])
    np.testing.assert_array_equal(idx, g[tag + "_idx"])
    np.testing.assert_array_equal(d2, g[tag + "_dist2"])                # lattice: exact in fp32
    np.testing.assert_allclose(np.sqrt(d2), g[tag + "_dist"], rtol=2.4e-7, atol=0)   # torch's CPU sqrt: <= 1 ulp off


def test_three_nn_and_interpolation():
    unknown = lattice_points("nn_u", (2, 20, 3))
    known = lattice_points("nn_k", (2, 11, 3))
    d2, idx = cref.nearestneighbor(unknown, known)
    full = ((unknown[:, :, None].astype(np.float64) - known[:, None]) ** 2).sum(-1)
    order = np.argsort(full, axis=2, kind="stable")[:, :, :3]
    np.testing.assert_array_equal(idx, order)
    np.testing.assert_allclose(d2, np.take_along_axis(full, order, 2), rtol=1e-6)
    feats = hash_tensor("nn_f", (2, 4, 11)).numpy()
    w = np.abs(hash_tensor("nn_w", (2, 20, 3)).numpy())
    out = cref.interpolation_forward(feats, idx, w)
    ref = sum(np.take_along_axis(feats, np.broadcast_to(idx[:, None, :, t], (2, 4, 20)), 2) * w[:, None, :, t]
              for t in range(3))
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)
    g = hash_tensor("nn_g", (2, 4, 20)).numpy()
    gb = cref.interpolation_backward(g, idx, w, 11)
    # <g, J f> == <J^T g, f>
    np.testing.assert_allclose((g * out).sum(), (gb * feats).sum(), rtol=1e-4)


# ------------------------------------------------------------------ structural losses
def test_nndistance_matches_reference_distchamfer(golden):
    g = golden("chamfer.npz")
    d1, i1, d2, i2 = cref.nndistance(g["a"], g["b"])
    np.testing.assert_allclose(d1, g["dist_r"], rtol=1e-4, atol=1e-5)   # min over b for each a
    np.testing.assert_allclose(d2, g["dist_l"], rtol=1e-4, atol=1e-5)
    full = ((g["a"][:, :, None].astype(np.float64) - g["b"][:, None]) ** 2).sum(-1)
    np.testing.assert_array_equal(i1, full.argmin(2))
    np.testing.assert_array_equal(i2, full.argmin(1))


def test_nndistance_grad_is_gradient_of_sum():
    a = hash_tensor("ndg_a", (2, 9, 3)).double().requires_grad_(True)
    b = hash_tensor("ndg_b", (2, 7, 3)).double().requires_grad_(True)
    P = ((a[:, :, None] - b[:, None]) ** 2).sum(-1)
    g1 = hash_tensor("ndg_g1", (2, 9)).double()
    g2 = hash_tensor("ndg_g2", (2, 7)).double()
    ((P.min(2)[0] * g1).sum() + (P.min(1)[0] * g2).sum()).backward()
    d1, i1, d2, i2 = cref.nndistance(a.detach().numpy(), b.detach().numpy())
    ga, gb = cref.nndistance_grad(a.detach().numpy(), b.detach().numpy(), i1, i2, g1.numpy(), g2.numpy())
    np.testing.assert_allclose(ga, a.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gb, b.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_emd_known_answers():
    """properties the approximation is known to satisfy."""
    n = 128
    a = hash_tensor("emd_a", (2, n, 3)).numpy()
    match = cref.approxmatch(a, a)
    np.testing.assert_allclose(cref.matchcost(a, a, match), 0.0, atol=1e-4)
    assert (np.abs(match.sum(1) - 1) < 1e-3).all() and (np.abs(match.sum(2) - 1) < 1e-3).all()
    assert (match[:, np.arange(n), np.arange(n)] > 0.99).all()
    t = np.array([0.004, -0.002, 0.001], np.float32)
    np.testing.assert_allclose(cref.emd_approx(a, a + t), np.linalg.norm(t), rtol=2e-2)
    perm = np.argsort(hash_tensor("emd_perm", (n,)).numpy())
    b = hash_tensor("emd_b", (2, n, 3), salt=1).numpy()
    np.testing.assert_allclose(cref.emd_approx(a[:, perm], b), cref.emd_approx(a, b), rtol=1e-4)
    m = cref.approxmatch(a, b)
    assert (m.sum(1) <= 1 + 1e-4).all() and (m.sum(2) <= 1 + 1e-4).all() and (m >= 0).all()


def test_matchcost_grad_is_gradient_for_fixed_match():
    a = hash_tensor("mcg_a", (1, 12, 3)).double().requires_grad_(True)
    b = hash_tensor("mcg_b", (1, 12, 3), salt=2).double().requires_grad_(True)
    match = cref.approxmatch(a.detach().numpy(), b.detach().numpy())
    d = ((b[:, :, None] - a[:, None]) ** 2).sum(-1).sqrt()          # (1, m, n)
    (torch.from_numpy(match).double() * d).sum().backward()
    g1, g2 = cref.matchcost_grad(a.detach().numpy(), b.detach().numpy(), match)
    np.testing.assert_allclose(g1, a.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g2, b.grad.numpy(), rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ torch restatement vs reference
def test_edge_features(golden):
    g = golden("edge_features.npz")
    x, pc = torch.from_numpy(g["x"]), torch.from_numpy(g["pc"])
    idx, dist = pdgnet_ref.feature_knn(x, int(g["k"]))
    np.testing.assert_array_equal(idx.numpy(), g["idx"])
    np.testing.assert_array_equal(pdgnet_ref.edge_features(x, idx).numpy(), g["e_fea"])
    np.testing.assert_array_equal(pdgnet_ref.edge_features(pc, idx).numpy(), g["e_xyz"])


@pytest.mark.parametrize("name", ["plain_k4", "bilateral_k4", "plain_k10", "bilateral_k10"])
def test_deconv_block(golden, name):
    g = golden("deconv_%s.npz" % name)
    bilateral = name.startswith("bilateral")
    mod = pdgnet_ref.EdgeDeconvRef(int(g["F"]), int(g["Fout"]), int(g["k"]), bilateral)
    fill_module(mod, salt=3)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    pc = torch.from_numpy(g["pc"]).requires_grad_(True) if bilateral else None
    idx = torch.from_numpy(g["idx"])
    mod.train()
    y = mod(x, pc, idx=idx)
    np.testing.assert_allclose(y.detach().numpy(), g["y_train"], rtol=1e-5, atol=1e-6)
    y.backward(torch.from_numpy(g["gout"]))
    np.testing.assert_allclose(x.grad.numpy(), g["grad_x"], rtol=1e-4, atol=1e-6)
    if bilateral:
        np.testing.assert_allclose(pc.grad.numpy(), g["grad_pc"], rtol=1e-4, atol=1e-6)
    for n, p in mod.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), g["grad." + n], rtol=1e-4, atol=1e-5, err_msg=n)
    for n, b in mod.named_buffers():
        if "num_batches" not in n:
            np.testing.assert_allclose(b.numpy(), g["stat." + n], rtol=1e-5, atol=1e-6, err_msg=n)
    mod.eval()
    with torch.no_grad():
        np.testing.assert_allclose(mod(x, pc).numpy(), g["y_eval"], rtol=1e-5, atol=1e-6)


def test_state_dict_manifest():
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        man = json.load(f)
    G = pdgnet_ref.PointGeneratorRef()
    assert {k: list(v.shape) for k, v in G.state_dict().items()} == man["G"]
    for i in (1, 2, 3, 4):
        D = pdgnet_ref.PointDiscriminatorRef(i)
        assert {k: list(v.shape) for k, v in D.state_dict().items()} == man["D%d" % i]


def test_generator_and_discriminators(golden):
    g = golden("generator_b6.npz")
    G = fill_module(pdgnet_ref.PointGeneratorRef(), salt=1).train()
    with torch.no_grad():
        outs = G(torch.from_numpy(g["z"]),
                 idx=[torch.from_numpy(g["idx%d" % i].astype(np.int64)) for i in (1, 2, 3, 4)])
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), g["p%d" % (i + 1)], rtol=1e-4, atol=1e-5)
    for i in (1, 2, 3, 4):
        D = fill_module(pdgnet_ref.PointDiscriminatorRef(i), salt=9 + i).train()
        with torch.no_grad():
            np.testing.assert_allclose(D(torch.from_numpy(g["p%d" % i])).numpy(), g["d%d" % i],
                                       rtol=1e-4, atol=1e-5)


def test_chamfer_and_local_stats(golden):
    g = golden("chamfer.npz")
    t = lambda k: torch.from_numpy(g[k])
    np.testing.assert_allclose(pdgnet_ref.chamfer_loss_sum(t("a"), t("b")).item(), g["chamfer_sum3"], rtol=1e-5)
    np.testing.assert_allclose(pdgnet_ref.chamfer_loss_sum(t("a9"), t("b9")).item(), g["chamfer_sum9"], rtol=1e-5)
    dl, dr = pdgnet_ref.dist_chamfer(t("a"), t("b"))
    np.testing.assert_allclose(dl.numpy(), g["dist_l"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dr.numpy(), g["dist_r"], rtol=1e-5, atol=1e-6)
    mu, cov = pdgnet_ref.mean_covariance(t("mc_points"))
    np.testing.assert_allclose(mu.numpy(), g["mc_mu"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(cov.numpy(), g["mc_cov"], rtol=1e-6, atol=1e-7)


def test_oracle_extra_pointops_known_answers():
    """the C restatements of the entry points PDGN never calls (parity unpinned): numpy brute force / properties"""
    rng = np.random.default_rng(21)
    xyz = rng.standard_normal((2, 120, 3)).astype(np.float32)
    q = rng.standard_normal((2, 30, 3)).astype(np.float32)
    r, ns = 0.8, 6
    idx = cref.ballquery(r, ns, xyz, q)
    d2 = ((q[:, :, None, :].astype(np.float64) - xyz[:, None, :, :]) ** 2).sum(-1)
    for b in range(2):
        for j in range(30):
            inside = np.nonzero(d2[b, j] < r * r)[0]
            if len(inside) == 0:
                assert (idx[b, j] == 0).all()
                continue
            k = min(ns, len(inside))
            np.testing.assert_array_equal(idx[b, j, :k], inside[:k])
            assert (idx[b, j, k:] == inside[0]).all()
    fps = cref.furthestsampling(xyz, 10)
    for b in range(2):
        sel, mind = [0], ((xyz[b] - xyz[b, 0]) ** 2).sum(1)
        for _ in range(9):
            nxt = int(mind.argmax())
            sel.append(nxt)
            mind = np.minimum(mind, ((xyz[b] - xyz[b, nxt]) ** 2).sum(1))
        np.testing.assert_array_equal(fps[b], sel)
    feat = rng.standard_normal((2, 4, 120)).astype(np.float32)
    gi = rng.integers(0, 120, (2, 50)).astype(np.int32)
    np.testing.assert_array_equal(cref.gathering_forward(feat, gi), np.take_along_axis(feat, gi[:, None, :].astype(np.int64), 2))
    g = rng.standard_normal((2, 4, 50)).astype(np.float32)
    want = np.zeros((2, 4, 120), np.float32)
    for b in range(2):
        np.add.at(want[b], (slice(None), gi[b]), g[b])
    np.testing.assert_allclose(cref.gathering_backward(g, gi, 120), want, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(cref.featuredistribute(xyz, q), d2.argmin(2).astype(np.int32))
    stat = np.zeros((2, 120, 5), np.int32)
    np.put_along_axis(stat, rng.integers(0, 5, (2, 120))[:, :, None], 1, 2)
    np.testing.assert_array_equal(cref.labelstat_ballrange(r, xyz, q, stat),
                                  np.einsum("bmn,bnc->bmc", (d2 < r * r).astype(np.int64), stat).astype(np.int32))
    st, bi = cref.labelstat_and_ballquery(r, ns, xyz, q, stat)
    np.testing.assert_array_equal(bi, idx)
    gathered = np.take_along_axis(stat, idx.reshape(2, -1)[:, :, None].astype(np.int64), 1).reshape(2, 30, ns, 5).sum(2)
    np.testing.assert_array_equal(cref.labelstat_idx(ns, stat, idx), gathered)
    cnt = np.minimum((d2 < r * r).sum(2), ns)
    np.testing.assert_array_equal(st.sum(2), cnt)
