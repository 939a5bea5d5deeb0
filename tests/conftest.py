import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("rtlsdr-airband_amd")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) libairband_hip.so and the C oracle; both are needed by most tests."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    ge.build()
    return True
