import os
import sys

import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
for p in (_HERE, _ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def nm():
    """The product binding; GPU tests fail loudly when libnm_hip.so is missing (no fallback)."""
    import niftymatch_amd
    niftymatch_amd.lib()
    return niftymatch_amd


@pytest.fixture(scope="session")
def cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test needs a GPU"
    return torch.device("cuda:0")
