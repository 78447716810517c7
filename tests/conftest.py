import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library binding; GPU tests only."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from gnn_fpga_amd import _lib
    return _lib


from gnn_fpga_amd.model import SegmentClassifier as _SegClf

DEFAULT_USE_PLAN = _SegClf.use_plan          # what a user gets ("auto"): pinned by test_abi_and_host / test_gpu_parity


@pytest.fixture(autouse=True)
def fused_route_from_the_first_forward(monkeypatch):
    """The parity tests exercise the fused tile pipeline (and inspect the plan) at the FIRST forward of a batch:
    inside tests SegmentClassifier.use_plan is True.  The default policy ("auto": per-module kernels for the first
    forward of a never-seen batch) has its own test, which sets it on the model."""
    monkeypatch.setattr(_SegClf, "use_plan", True)
