"""numpy binding of the C oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Every function takes/returns numpy arrays with the reference's layouts; see the
C sources for the reference file:line each one restates.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("pointops_ref.c", "structural_ref.c", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def knnquery(nsample, xyz, new_xyz=None):
    xyz = _f32(xyz)
    new_xyz = xyz if new_xyz is None else _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    dist2 = np.zeros((b, m, nsample), np.float32)
    rc = lib().oracle_knnquery(b, n, m, nsample, _p(xyz), _p(new_xyz), _p(idx), _p(dist2))
    if rc:
        raise RuntimeError("oracle_knnquery: nsample out of range")
    return idx, dist2


def grouping_forward(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, c, m, ns), np.float32)
    lib().oracle_grouping_forward(b, c, n, m, ns, _p(points), _p(idx), _p(out))
    return out


def grouping_backward(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m, ns = grad_out.shape
    g = np.zeros((b, c, n), np.float32)
    lib().oracle_grouping_backward(b, c, n, m, ns, _p(grad_out), _p(idx), _p(g))
    return g


def nearestneighbor(unknown, known):
    unknown, known = _f32(unknown), _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    lib().oracle_nearestneighbor(b, n, m, _p(unknown), _p(known), _p(dist2), _p(idx))
    return dist2, idx


def interpolation_forward(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.empty((b, c, n), np.float32)
    lib().oracle_interpolation_forward(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def interpolation_backward(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    b, c, n = grad_out.shape
    g = np.zeros((b, c, m), np.float32)
    lib().oracle_interpolation_backward(b, c, n, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


def nndistance(xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.empty((b, n), np.float32)
    i1 = np.empty((b, n), np.int32)
    d2 = np.empty((b, m), np.float32)
    i2 = np.empty((b, m), np.int32)
    lib().oracle_nndistance(b, n, _p(xyz1), m, _p(xyz2), _p(d1), _p(i1), _p(d2), _p(i2))
    return d1, i1, d2, i2


def nndistance_grad(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    idx1, idx2 = _i32(idx1), _i32(idx2)
    grad_dist1, grad_dist2 = _f32(grad_dist1), _f32(grad_dist2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.empty((b, n, 3), np.float32)
    g2 = np.empty((b, m, 3), np.float32)
    lib().oracle_nndistance_grad(b, n, _p(xyz1), m, _p(xyz2), _p(grad_dist1), _p(idx1),
                                 _p(grad_dist2), _p(idx2), _p(g1), _p(g2))
    return g1, g2


def approxmatch(xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.empty((b, m, n), np.float32)
    temp = np.empty((b, 2 * (n + m)), np.float32)
    lib().oracle_approxmatch(b, n, m, _p(xyz1), _p(xyz2), _p(match), _p(temp))
    return match


def matchcost(xyz1, xyz2, match):
    xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    out = np.empty((b,), np.float32)
    lib().oracle_matchcost(b, n, m, _p(xyz1), _p(xyz2), _p(match), _p(out))
    return out


def matchcost_grad(xyz1, xyz2, match):
    xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.empty((b, n, 3), np.float32)
    g2 = np.empty((b, m, 3), np.float32)
    lib().oracle_matchcost_grad(b, n, m, _p(xyz1), _p(xyz2), _p(match), _p(g1), _p(g2))
    return g1, g2


def emd_approx(xyz1, xyz2):
    """evaluation/evaluation_metrics.py:26-31: match_cost / N."""
    match = approxmatch(xyz1, xyz2)
    return matchcost(xyz1, xyz2, match) / np.float32(xyz1.shape[1])


# ---- entry points PDGN never calls (parity unpinned: no Python twin / test in the reference) ----
def ballquery(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().oracle_ballquery(b, n, m, ctypes.c_float(radius), nsample, _p(new_xyz), _p(xyz), _p(idx))
    return idx


def furthestsampling(xyz, m):
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32)
    idx = np.zeros((b, m), np.int32)
    lib().oracle_furthestsampling(b, n, m, _p(xyz), _p(temp), _p(idx))
    return idx


def gathering_forward(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.empty((b, c, m), np.float32)
    lib().oracle_gathering_forward(b, c, n, m, _p(points), _p(idx), _p(out))
    return out


def gathering_backward(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m = grad_out.shape
    g = np.zeros((b, c, n), np.float32)
    lib().oracle_gathering_backward(b, c, n, m, _p(grad_out), _p(idx), _p(g))
    return g


def grouping_int_forward(points, idx):
    points = np.ascontiguousarray(points, dtype=np.int64)
    idx = _i32(idx)
    b, c, n = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, c, m, ns), np.int64)
    lib().oracle_grouping_int_forward(b, c, n, m, ns, _p(points), _p(idx), _p(out))
    return out


def featuredistribute(max_xyz, xyz):
    max_xyz, xyz = _f32(max_xyz), _f32(xyz)
    b, n, _ = max_xyz.shape
    m = xyz.shape[1]
    out = np.empty((b, m), np.int32)
    lib().oracle_featuredistribute(b, n, m, _p(max_xyz), _p(xyz), _p(out))
    return out


def labelstat_idx(nsample, label_stat, idx):
    label_stat, idx = _i32(label_stat), _i32(idx)
    b, n, nclass = label_stat.shape
    m = idx.shape[1]
    out = np.empty((b, m, nclass), np.int32)
    lib().oracle_labelstat_idx(b, n, m, nsample, nclass, _p(label_stat), _p(idx), _p(out))
    return out


def labelstat_ballrange(radius, xyz, new_xyz, label_stat):
    xyz, new_xyz, label_stat = _f32(xyz), _f32(new_xyz), _i32(label_stat)
    b, n, nclass = label_stat.shape
    m = new_xyz.shape[1]
    out = np.empty((b, m, nclass), np.int32)
    lib().oracle_labelstat_ball(b, n, m, ctypes.c_float(radius), 0, nclass, 0, _p(new_xyz), _p(xyz), _p(label_stat),
                                None, _p(out))
    return out


def labelstat_and_ballquery(radius, nsample, xyz, new_xyz, label_stat):
    xyz, new_xyz, label_stat = _f32(xyz), _f32(new_xyz), _i32(label_stat)
    b, n, nclass = label_stat.shape
    m = new_xyz.shape[1]
    out = np.empty((b, m, nclass), np.int32)
    idx = np.zeros((b, m, nsample), np.int32)
    lib().oracle_labelstat_ball(b, n, m, ctypes.c_float(radius), nsample, nclass, 1, _p(new_xyz), _p(xyz),
                                _p(label_stat), _p(idx), _p(out))
    return out, idx
