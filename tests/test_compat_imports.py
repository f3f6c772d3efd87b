"""The reference's own import lines (models/PDGNet_v2.py:17, evaluation/evaluation_metrics.py:9-10) resolve to
pdgn_amd when compat/ is on sys.path ("same import paths & names")."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "compat")

# module-level names of lib/pointops/functions/pointops.py:30-777
POINTOPS_NAMES = ("furthestsampling gathering nearestneighbor interpolation grouping grouping_int ballquery "
                  "featuredistribute featuregather labelstat_ballrange labelstat_idx labelstat_and_ballquery "
                  "knnquery_naive knnquery knnquery_exclude pairwise_distances QueryAndGroup QueryAndGroup_Dilate "
                  "Le_QueryAndGroup Le_QueryAndGroup_SameSize Le_QueryAndGroup_OnlyFeature Gen_QueryAndGroupXYZ GroupAll "
                  "KNNQuery Grouping NearestNeighbor Interpolation Gathering").split()


@pytest.fixture()
def compat_path():
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    for name in [m for m in sys.modules if m == "lib" or m.startswith("lib.") or m == "evaluation" or m.startswith("evaluation.")]:
        del sys.modules[name]
    sys.path.insert(0, COMPAT)
    yield
    sys.path[:] = saved_path
    for name in [m for m in sys.modules if m not in saved_mods]:
        del sys.modules[name]


def test_reference_import_lines_resolve_to_pdgn_amd(compat_path):
    from lib.pointops.functions import pointops                       # models/PDGNet_v2.py:17
    from evaluation.StructuralLosses.match_cost import match_cost     # evaluation/evaluation_metrics.py:9
    from evaluation.StructuralLosses.nn_distance import nn_distance   # evaluation/evaluation_metrics.py:10
    import pdgn_amd.pointops
    import pdgn_amd.structural_losses as sl
    for name in POINTOPS_NAMES:
        assert getattr(pointops, name) is getattr(pdgn_amd.pointops, name), name
    assert match_cost is sl.match_cost and nn_distance is sl.nn_distance


@pytest.mark.skipif(not os.path.isdir("/root/reference/evaluation"), reason="reference checkout not on this machine")
def test_reference_modules_import_through_compat():
    """The reference's evaluation_metrics.py and models/PDGNet_v2.py imported UNEDITED, compat/ in front of the reference on
    sys.path (a child process: the reference wants `h5py`, absent here, which gets an empty stand-in module)."""
    code = (
        "import sys, types; sys.modules['h5py'] = types.ModuleType('h5py');"
        "sys.path[:0] = [%r, %r, '/root/reference'];"
        "import evaluation.evaluation_metrics as em, models.PDGNet_v2 as net, pdgn_amd.pointops as po, pdgn_amd.structural_losses as sl;"
        "assert em.match_cost is sl.match_cost and em.nn_distance is sl.nn_distance;"
        "assert net.pointops.knnquery is po.knnquery and net.pointops.Gen_QueryAndGroupXYZ is po.Gen_QueryAndGroupXYZ;"
        "assert em.__file__.startswith('/root/reference'); print('ok')" % (COMPAT, ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


@pytest.mark.gpu
def test_calls_through_the_reference_import_paths(compat_path):
    """Gen_QueryAndGroupXYZ / match_cost / nn_distance called through the reference's import lines, checked against the oracle."""
    from lib.pointops.functions import pointops
    from evaluation.StructuralLosses.match_cost import match_cost
    from evaluation.StructuralLosses.nn_distance import nn_distance
    from oracle import cref
    rng = np.random.default_rng(11)
    xyz = rng.standard_normal((2, 300, 3)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :64])
    grouper = pointops.Gen_QueryAndGroupXYZ(radius=None, nsample=20, use_xyz=False)       # models/PDGNet_v2.py:139
    out = grouper(xyz=torch.from_numpy(xyz).cuda(), new_xyz=torch.from_numpy(new_xyz).cuda())
    idx, _ = cref.knnquery(20, xyz, new_xyz)
    np.testing.assert_array_equal(out.cpu().numpy(), cref.grouping_forward(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx))
    a = rng.uniform(-1, 1, (2, 256, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (2, 256, 3)).astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    np.testing.assert_allclose(match_cost(ta, tb).cpu().numpy(), cref.matchcost(a, b, cref.approxmatch(a, b)), rtol=1e-4)
    d1, d2 = nn_distance(ta, tb)
    r1, _, r2, _ = cref.nndistance(a, b)
    np.testing.assert_array_equal(d1.cpu().numpy(), r1)
    np.testing.assert_array_equal(d2.cpu().numpy(), r2)
