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


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment(request):
    """The library reads its environment knobs once; a test that flips one (monkeypatch.setenv + leann_debug_reload_env) must not leak it
    into the next test: after the test — and after monkeypatch has restored the environment (autouse fixtures are torn down last) — the
    knobs are read again."""
    yield
    if "la" in request.fixturenames:
        request.getfixturevalue("la").lib().leann_debug_reload_env()
