"""The pointops entry points PDGN never calls: HIP vs the C restatement
(bit-exact: integer outputs) plus known-answer properties.  Parity with the reference is UNPINNED for these
(CUDA-only there, no Python twin): the oracle follows the cited kernels line by line."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _clouds(seed, b, n, m):
    rng = np.random.default_rng(seed)
    xyz = rng.standard_normal((b, n, 3)).astype(np.float32)
    q = rng.standard_normal((b, m, 3)).astype(np.float32)
    return xyz, q


@pytest.mark.parametrize("b,n,m,ns,r", [(2, 300, 70, 8, 0.6), (1, 2500, 300, 32, 0.35), (3, 17, 5, 4, 0.05), (2, 64, 64, 16, 10.0)])
def test_ballquery(b, n, m, ns, r):
    from pdgn_amd import pointops as po
    xyz, q = _clouds(n + m, b, n, m)
    got = po.ballquery(r, ns, dev(xyz), dev(q)).cpu().numpy()
    np.testing.assert_array_equal(got, cref.ballquery(r, ns, xyz, q))
    d2 = ((q[:, :, None, :].astype(np.float64) - xyz[:, None, :, :]) ** 2).sum(-1)
    for bi in range(b):
        for j in range(m):
            inside = np.nonzero(d2[bi, j] < r * r * (1 - 1e-6))[0]
            if len(inside) == 0:
                continue
            k = min(ns, len(inside))
            if (np.abs(d2[bi, j] - r * r) > 1e-5).all():               # no point on the sphere: order is unambiguous
                np.testing.assert_array_equal(got[bi, j, :k], inside[:k])
                assert (got[bi, j, k:] == inside[0]).all()


@pytest.mark.parametrize("b,n,m", [(2, 500, 64), (1, 2048, 256), (3, 40, 40), (1, 5000, 16)])
def test_furthestsampling(b, n, m):
    from pdgn_amd import pointops as po
    xyz, _ = _clouds(n, b, n, 1)
    got = po.furthestsampling(dev(xyz), m).cpu().numpy()
    np.testing.assert_array_equal(got, cref.furthestsampling(xyz, m))
    assert (got[:, 0] == 0).all()
    for bi in range(b):
        assert len(set(got[bi].tolist())) == m                        # distinct while m <= n (random data)
        sel = xyz[bi, got[bi, :2]].astype(np.float64)                  # second pick = farthest from point 0
        d0 = ((xyz[bi].astype(np.float64) - sel[0]) ** 2).sum(1)
        assert abs(d0.max() - ((sel[1] - sel[0]) ** 2).sum()) < 1e-6


def test_gathering_and_featuregather_with_grad():
    from pdgn_amd import pointops as po
    rng = np.random.default_rng(5)
    f = rng.standard_normal((3, 7, 90)).astype(np.float32)
    idx = rng.integers(0, 90, (3, 200)).astype(np.int32)
    g = rng.standard_normal((3, 7, 200)).astype(np.float32)
    fd = dev(f).requires_grad_(True)
    out = po.gathering(fd, dev(idx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), cref.gathering_forward(f, idx))
    out.backward(dev(g))
    np.testing.assert_allclose(fd.grad.cpu().numpy(), cref.gathering_backward(g, idx, 90), rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(po.featuregather(dev(f), dev(idx)).cpu().numpy(), cref.gathering_forward(f, idx))


def test_grouping_int():
    from pdgn_amd import pointops as po
    rng = np.random.default_rng(6)
    f = rng.integers(-2 ** 40, 2 ** 40, (2, 3, 50)).astype(np.int64)
    idx = rng.integers(0, 50, (2, 20, 6)).astype(np.int32)
    got = po.grouping_int(dev(f), dev(idx)).cpu().numpy()
    np.testing.assert_array_equal(got, cref.grouping_int_forward(f, idx))
    np.testing.assert_array_equal(got[1, 2, 5, 3], f[1, 2, idx[1, 5, 3]])


@pytest.mark.parametrize("b,n,m", [(2, 33, 400), (1, 1500, 1500)])
def test_featuredistribute(b, n, m):
    from pdgn_amd import pointops as po
    mx, xyz = _clouds(n * 3 + m, b, n, m)
    got = po.featuredistribute(dev(mx), dev(xyz)).cpu().numpy()
    np.testing.assert_array_equal(got, cref.featuredistribute(mx, xyz))
    d2 = ((xyz[:, :, None, :].astype(np.float64) - mx[:, None, :, :]) ** 2).sum(-1)
    assert (np.take_along_axis(d2, got[:, :, None].astype(np.int64), 2)[:, :, 0] <= d2.min(2) + 1e-6).all()


def test_labelstat_family():
    from pdgn_amd import pointops as po
    rng = np.random.default_rng(8)
    b, n, m, ns, nclass = 2, 400, 90, 12, 13
    xyz, q = _clouds(77, b, n, m)
    label = rng.integers(0, nclass, (b, n))
    stat = np.zeros((b, n, nclass), np.int32)
    np.put_along_axis(stat, label[:, :, None], 1, 2)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    np.testing.assert_array_equal(po.labelstat_idx(ns, dev(stat), dev(idx)).cpu().numpy(), cref.labelstat_idx(ns, stat, idx))
    got = po.labelstat_ballrange(0.7, dev(xyz), dev(q), dev(stat)).cpu().numpy()
    np.testing.assert_array_equal(got, cref.labelstat_ballrange(0.7, xyz, q, stat))
    d2 = ((q[:, :, None, :] - xyz[:, None, :, :]) ** 2).sum(-1)
    assert abs(int(got.sum()) - int((d2 < 0.49).sum())) <= 2            # one-hot labels: histogram mass = ball population
    st, bi = po.labelstat_and_ballquery(0.7, ns, dev(xyz), dev(q), dev(stat))
    want_st, want_idx = cref.labelstat_and_ballquery(0.7, ns, xyz, q, stat)
    np.testing.assert_array_equal(st.cpu().numpy(), want_st)
    np.testing.assert_array_equal(bi.cpu().numpy(), want_idx)
    np.testing.assert_array_equal(bi.cpu().numpy(), po.ballquery(0.7, ns, dev(xyz), dev(q)).cpu().numpy())
    assert (st.cpu().numpy().sum(2) <= ns).all()


def test_query_and_group_with_radius_uses_ballquery():
    """QueryAndGroup(radius=...) (pointops.py:476-540) routes through the HIP ball query"""
    from pdgn_amd import pointops as po
    xyz, _ = _clouds(3, 2, 128, 1)
    feat = np.random.default_rng(4).standard_normal((2, 5, 128)).astype(np.float32)
    mod = po.QueryAndGroup(radius=0.9, nsample=8, use_xyz=True)
    out = mod(dev(xyz), dev(xyz[:, :32].copy()), dev(feat))
    assert out.shape == (2, 8, 32, 8)
    idx = cref.ballquery(0.9, 8, xyz, xyz[:, :32])
    want = cref.grouping_forward(feat, idx)
    np.testing.assert_allclose(out[:, 3:].cpu().numpy(), want, rtol=0, atol=0)
