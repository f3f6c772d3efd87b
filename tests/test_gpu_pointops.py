"""GPU parity: pdgn_amd.pointops (HIP, through the C ABI) vs the C oracle and the golden vectors.
Index outputs are compared bit-exactly."""
import numpy as np
import pytest
import torch

from hashweights import hash_tensor, lattice_points
from oracle import cref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def po():
    from pdgn_amd import pointops
    return pointops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_knnquery_golden(po, golden, tag):
    g = golden("pointops_knn.npz")
    idx = po.knnquery(int(g[tag + "_k"]), dev(g[tag + "_xyz"]), dev(g[tag + "_new_xyz"]))
    assert idx.dtype == torch.int32
    np.testing.assert_array_equal(idx.cpu().numpy(), g[tag + "_idx"])


@pytest.mark.parametrize("b,n,m,k", [(2, 256, 256, 20), (3, 2048, 256, 20), (2, 1024, 1024, 20),
                                     (1, 5000, 77, 20), (2, 300, 50, 1), (2, 70, 33, 32),
                                     (1, 333, 40, 48), (1, 500, 9, 200), (2, 7, 5, 20), (1, 1, 1, 3)])
def test_knnquery_vs_oracle_bitexact(po, b, n, m, k):
    rng = np.random.default_rng(n * 7 + m)
    xyz = rng.standard_normal((b, n, 3)).astype(np.float32)
    q = rng.standard_normal((b, m, 3)).astype(np.float32)
    if m <= n:
        q[:, : m // 2] = xyz[:, : m // 2]            # queries that are members of the set
    ref_idx, ref_d = cref.knnquery(k, xyz, q)
    idx, d2 = po.knnquery_with_dist(k, dev(xyz), dev(q))
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx)
    np.testing.assert_array_equal(d2.cpu().numpy(), ref_d)


def test_knnquery_duplicates_and_self_default(po):
    xyz = np.zeros((2, 130, 3), np.float32)
    xyz[:, 65:] = 1.0
    xyz[1, ::3] = 0.5
    ref_idx, _ = cref.knnquery(20, xyz, xyz)
    idx = po.knnquery(20, dev(xyz))                  # new_xyz=None => xyz (pointops.py:418-419)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx)


def test_knnquery_survivor_overflow(po):
    # > 256 candidates at exactly the same distance force mid-scan queue flushes
    xyz = np.zeros((1, 1500, 3), np.float32)
    xyz[0, :, 0] = np.where(np.arange(1500) % 5 == 0, 2.0, 1.0)
    q = np.zeros((1, 3, 3), np.float32)
    ref_idx, ref_d = cref.knnquery(20, xyz, q)
    idx, d2 = po.knnquery_with_dist(20, dev(xyz), dev(q))
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx)
    np.testing.assert_array_equal(d2.cpu().numpy(), ref_d)


@pytest.mark.parametrize("b,c,n,m,ns", [(2, 3, 256, 256, 20), (2, 3, 2048, 1024, 20), (2, 64, 100, 37, 5),
                                        (1, 5, 20000, 64, 8), (1, 1, 1, 1, 1)])
def test_grouping_forward_backward(po, b, c, n, m, ns):
    rng = np.random.default_rng(5)
    feats = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    idx[:, :, 0] = 0                                  # a hot destination
    f = dev(feats).requires_grad_(True)
    out = po.grouping(f, dev(idx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), cref.grouping_forward(feats, idx))
    g = rng.standard_normal((b, c, m, ns)).astype(np.float32)
    out.backward(dev(g))
    np.testing.assert_allclose(f.grad.cpu().numpy(), cref.grouping_backward(g, idx, n), rtol=1e-4, atol=1e-4)


def test_gen_query_and_group_xyz(po):
    rng = np.random.default_rng(11)
    xyz = rng.standard_normal((2, 512, 3)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :256])
    grp = po.Gen_QueryAndGroupXYZ(radius=None, nsample=20, use_xyz=False)
    out = grp(dev(xyz), dev(new_xyz))
    idx, _ = cref.knnquery(20, xyz, new_xyz)
    ref = cref.grouping_forward(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx)
    assert out.shape == (2, 3, 256, 20)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_nearestneighbor_golden(po, golden, tag):
    """The reference's own answer (KNNQueryNaive with nsample 3 on lattice inputs, gen_golden.py::gen_nn3)."""
    g = golden("pointops_nn3.npz")
    dist, idx = po.nearestneighbor(dev(g[tag + "_unknown"]), dev(g[tag + "_known"]))
    assert idx.dtype == torch.int32
    np.testing.assert_array_equal(idx.cpu().numpy(), g[tag + "_idx"])
    np.testing.assert_array_equal(dist.cpu().numpy(), np.sqrt(g[tag + "_dist2"]))   # exact squares, rounded sqrt
    np.testing.assert_allclose(dist.cpu().numpy(), g[tag + "_dist"], rtol=2.4e-7, atol=0)


@pytest.mark.parametrize("b,n,m", [(2, 300, 100), (1, 2048, 512), (2, 5, 2), (1, 64, 3000)])
def test_nearestneighbor(po, b, n, m):
    rng = np.random.default_rng(3)
    unknown = rng.standard_normal((b, n, 3)).astype(np.float32)
    known = rng.standard_normal((b, m, 3)).astype(np.float32)
    d2, idx = cref.nearestneighbor(unknown, known)
    dist, gi = po.nearestneighbor(dev(unknown), dev(known))
    np.testing.assert_array_equal(gi.cpu().numpy(), idx)
    np.testing.assert_array_equal(dist.cpu().numpy(), np.sqrt(d2))


@pytest.mark.parametrize("b,c,m,n", [(2, 16, 100, 300), (1, 3, 512, 2048), (1, 2, 20000, 50)])
def test_interpolation_forward_backward(po, b, c, m, n):
    rng = np.random.default_rng(4)
    feats = rng.standard_normal((b, c, m)).astype(np.float32)
    idx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    w = rng.random((b, n, 3)).astype(np.float32)
    f = dev(feats).requires_grad_(True)
    out = po.interpolation(f, dev(idx), dev(w))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), cref.interpolation_forward(feats, idx, w))
    g = rng.standard_normal((b, c, n)).astype(np.float32)
    out.backward(dev(g))
    np.testing.assert_allclose(f.grad.cpu().numpy(), cref.interpolation_backward(g, idx, w, m),
                               rtol=1e-4, atol=1e-4)


def test_argument_validation(po):
    from pdgn_amd._lib import PdgnHipError
    xyz = torch.zeros(1, 8, 3, device="cuda")
    with pytest.raises(PdgnHipError):
        po.knnquery(201, xyz, xyz)                    # > 200 (knnquery_cuda_kernel.cu:21-22)
    with pytest.raises(AssertionError):
        po.knnquery(4, torch.zeros(1, 3, 8, device="cuda").transpose(1, 2), xyz)
    with pytest.raises(TypeError):
        po.grouping(torch.zeros(1, 3, 8, device="cuda"), torch.zeros(1, 2, 2, dtype=torch.int64, device="cuda"))


def test_torch_side_api_members(po):
    rng = np.random.default_rng(9)
    xyz = dev(rng.standard_normal((2, 200, 3)).astype(np.float32))
    q = xyz[:, :50].contiguous()
    idx = po.knnquery(8, xyz, q)
    naive = po.knnquery_naive(8, xyz, q)
    assert (idx == naive).float().mean() > 0.999
    assert (po.knnquery_exclude(7, xyz, q) == naive[:, :, 1:]).float().mean() > 0.999
    bq = po.ballquery(0.5, 8, xyz, q)
    d2 = (q.unsqueeze(2) - xyz.unsqueeze(1)).pow(2).sum(3)
    picked = torch.gather(d2, 2, bq.long())
    assert (picked[:, :, 0] < 0.25).all()             # every query is a member => non-empty ball
    fps = po.furthestsampling(xyz, 16)
    assert fps.shape == (2, 16) and (fps[:, 0] == 0).all() and len(set(fps[0].tolist())) == 16
