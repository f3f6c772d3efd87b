import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))
    return load


@pytest.fixture(autouse=True)
def _reset_gemm_switches(request):
    """GPU tests select the contraction kernel / tile configuration through pdgn_amd._lib.set_gemm_mode / set_gemm_config
    (process-wide switches of libpdgn_hip.so): put the defaults back after every such test."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        from pdgn_amd import _lib
        _lib.set_gemm_mode(_lib.DEFAULT_GEMM_MODE)
        _lib.set_gemm_config(int(os.environ["PDGN_NT_CFG"]) if os.environ.get("PDGN_NT_CFG") else None)
