"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/pdgn_hip.h
declares; the product refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pdgn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(pdgn_\w+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from pdgn_amd import build
    so = build.build()
    handle = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 13
    for name in names:
        assert hasattr(handle, name), "missing export: " + name
    from pdgn_amd import _lib
    assert handle.pdgn_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback():
    from pdgn_amd import pointops
    from pdgn_amd._lib import PdgnHipError
    xyz = torch.zeros(1, 8, 3)
    with pytest.raises(PdgnHipError):
        pointops.knnquery(4, xyz, xyz)
    with pytest.raises(PdgnHipError):
        pointops.grouping(torch.zeros(1, 3, 8), torch.zeros(1, 2, 2, dtype=torch.int32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pdgn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_host_side_rules_of_the_contractions():
    """The host-only queries of the dense contractions (no device work: they run here): which arithmetic a call takes, which planes
    to make, how much workspace its tail wants -- the rules DESIGN.md section 4 states, and their mutual consistency."""
    from pdgn_amd import build
    L = ctypes.CDLL(build.build())
    L.pdgn_gemm_tail_workspace_floats.restype = ctypes.c_longlong
    L.pdgn_gemm_nt_ps_workspace_floats.restype = ctypes.c_longlong
    L.pdgn_gemm_tn_big_workspace_floats.restype = ctypes.c_longlong
    ll = ctypes.c_longlong
    old = L.pdgn_gemm_set_mode(2)
    try:
        # two parts for unsplit operands: the 256 x 128 tile, k >= 128, >= 20 GFLOP, <= 4.5 B to scan per kflop
        assert L.pdgn_gemm_two_part(ll(35840), 512, 5120, ll(35840 * 5120 * 4)) == 1
        assert L.pdgn_gemm_two_part(ll(35840), 12832, 128, ll(0)) == 1
        assert L.pdgn_gemm_two_part(ll(35840), 128, 12832, ll(35840 * 12832 * 4)) == 0       # 1.8 GB to scan for 118 GFLOP
        assert L.pdgn_gemm_two_part(ll(358400), 512, 64, ll(0)) == 0 and L.pdgn_gemm_two_part(ll(3000), 256, 8, ll(0)) == 0
        # ... for pre-split planes: from ~2 GFLOP on
        assert L.pdgn_gemm_two_part_planes(ll(17920), 256, 2560, ll(0)) == 1 and L.pdgn_gemm_two_part_planes(ll(17920), 6432, 64, ll(17920 * 64 * 4)) == 1
        assert L.pdgn_gemm_two_part_planes(ll(17920), 256, 2560, ll(17920 * 2560 * 4)) == 0 and L.pdgn_gemm_two_part_planes(ll(200), 256, 2560, ll(0)) == 0
        # workspaces: three-part planes = the unsplit call's; two-part planes = the 256 x 128 tile's; none with statistics
        for m, n, k in ((35840, 512, 5120), (17920, 256, 2560), (35840, 128, 12832), (4100, 132, 260)):
            assert L.pdgn_gemm_nt_ps_workspace_floats(ll(m), n, k, 3, 0) == L.pdgn_gemm_tail_workspace_floats(ll(m), n, k, 0)
            assert L.pdgn_gemm_nt_ps_workspace_floats(ll(m), n, k, 2, 1) == 0 and L.pdgn_gemm_tail_workspace_floats(ll(m), n, k, 1) == 0
            assert L.pdgn_gemm_nt_ps_workspace_floats(ll(m), n, k, 2, 0) % (256 * 128) == 0
        assert L.pdgn_gemm_nt_ps_workspace_floats(ll(35840), 128, 12832, 2, 0) > 0             # 140 tiles on 256 CUs: the flattened tail
        assert L.pdgn_gemm_tn_big_workspace_floats(ll(35840), 512, 5120) > 0 and L.pdgn_gemm_tn_big_workspace_floats(ll(35840), 512, 5120) % (256 * 128) == 0
        L.pdgn_gemm_set_mode(1)
        assert L.pdgn_gemm_two_part(ll(35840), 512, 5120, ll(0)) == 0 and L.pdgn_gemm_two_part_planes(ll(17920), 256, 2560, ll(0)) == 0
        L.pdgn_gemm_set_mode(0)
        assert L.pdgn_gemm_tail_workspace_floats(ll(35840), 512, 5120, 0) == 0 and L.pdgn_gemm_tn_big_workspace_floats(ll(35840), 512, 5120) == 0
    finally:
        L.pdgn_gemm_set_mode(old)
