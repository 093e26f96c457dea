import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ap():
    import admm_project_amd
    return admm_project_amd


@pytest.fixture(scope="session")
def gpu(ap):
    """The HIP engine on a real device; fails loudly (never skips silently to a CPU path)."""
    ap._lib.load()
    if ap._lib.device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests need the GPU box")
    return ap
