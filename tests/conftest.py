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


# GPU cases that run on the wavefront-FFT channelizer (csrc/channelizer_fft.hip) go to the END of a `-m gpu` run.  Round 3's last session rewrote that kernel
# without a GPU (checked on the host emulation only), and the driver runs the GPU suite with `-x`: ordered like this, the run first says
# everything about the path the benchmark measures, and then what the GPU makes of the rewritten kernel -- instead of stopping at the first FFT case and
# saying nothing about the rest.
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
