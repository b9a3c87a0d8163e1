import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def po():
    import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def la():
    """The product package; loading fails loudly if csrc/libleann_hip.so is missing."""
    import leann_rs_amd
    leann_rs_amd.lib()
    return leann_rs_amd


@pytest.fixture(scope="session")
def gpu(la):
    if la.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box")
    return 0
