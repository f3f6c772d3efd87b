"""CPU: the register accounting of every kernel INSIDE the built libpdgn_hip.so (AMDGPU metadata notes of its code objects, the
scan of tools/spill_table.py).  VERDICT r4 #2: no instance that the default path can launch may spill vector registers."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _x3_class(name):
    """(tile index, class, MS) of a gemm_x3_kernel instance name, as pdgn_gemm_set_shape's mask counts them."""
    m = re.search(r"gemm_x3_kernel<(.*)>", name)
    a = [t.strip() for t in m.group(1).split(",")]
    tm, tn, wm, wn, occ = (int(v) for v in a[:5])
    atomic, wt, at, epi, pw = (v == "true" for v in a[5:10])
    ms = int(a[10]) if len(a) > 10 else 32
    tile = {(4, 2, 2, 1): 0, (2, 2, 4, 1): 0, (2, 2, 2, 1): 1, (2, 1, 2, 2): 2}[(tm, tn, wm, occ)]      # (the 256 x 128 tile: four waves of 128 x 64 or eight of 64 x 64)
    return tile, (3 if epi else 2 if atomic else 1 if pw else 0), ms


@pytest.mark.timeout(600)
def test_no_launched_kernel_spills_vector_registers():
    import ctypes
    import spill_table
    from pdgn_amd import build as hip_build
    rows = spill_table.scan_built(hip_build.build())
    assert len(rows) > 150, "the scan saw %d kernels" % len(rows)
    mask = ctypes.CDLL(hip_build.build()).pdgn_gemm_set_shape(0)       # the process default: which classes run on 16x16x32
    bad, arm_only = [], []
    for r in rows:
        if r.get("vgpr_spill", 0) == 0:
            continue
        if "gemm_x3_kernel" in r["name"]:
            tile, cls, ms = _x3_class(r["name"])
            launched = (ms == 16) == bool((mask >> (4 * tile + cls)) & 1)
            (bad if launched else arm_only).append((r["name"], r["vgpr_spill"]))
        elif "gemm_nt_kernel" in r["name"]:
            arm_only.append((r["name"], r["vgpr_spill"]))                # PDGN_GEMM=fp32: the A/B arm and second opinion, not the step
        else:
            bad.append((r["name"], r["vgpr_spill"]))
    assert not bad, "kernels on the default path with spilled vector registers: %s" % bad
    # the measurement arms are listed, not hidden
    print("A/B-arm instances with spills (not launched by default):", arm_only)
