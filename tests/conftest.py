import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_err(a, b):
    """||a-b||_2 / ||b||_2 in float64 (b is the reference side)."""
    import torch
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    den = b.norm().item()
    return (a - b).norm().item() / (den if den > 0 else 1.0)


@pytest.fixture(scope="session")
def have_reference():
    return os.path.isdir("/root/reference/multimae")
