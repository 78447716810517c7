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
