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


# GPU cases on the side paths (the wavefront-FFT channelizer, CF32) go to the END of a `-m gpu` run: the driver runs the suite with `-x`, and ordered like this a run
# first says everything about the path the benchmark measures.  (Round 3 introduced the ordering for a kernel written without a GPU; round 4's first GPU call validated
# that kernel -- profiles/r04_experiments.md A -- and the ordering stayed as a convention.)
_FFT_PATH = ("fft_wave64", "SFMT_F32", "test_fft_channelizer_lds_budget", "test_gpu_wavefront_fft")  # (AFC on the matrix-core path: its one-hop spectrum launch stays on the shuffle kernel)


def pytest_collection_modifyitems(config, items):
    last = [it for it in items if it.get_closest_marker("gpu") and any(k in it.nodeid for k in _FFT_PATH)]
    if last:
        keep = [it for it in items if it not in last]
        items[:] = keep + last


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
