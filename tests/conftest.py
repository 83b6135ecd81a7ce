import os
import sys

import pytest

try:                      # PyTorch-ROCm wheels bundle their own HIP runtime; a process must not end up with two of them (the
    import torch  # noqa: F401   second one finds no GPU), so torch's is loaded before libmi355cd.so pulls in /opt/rocm's
except Exception:         # (tests that need torch skip or fail on their own)
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gpu-computing-course_amd")
for p in (ROOT, os.path.join(PKG, "pyhost"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle and the HIP libraries exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    ge.build(quiet=True)
    yield


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
