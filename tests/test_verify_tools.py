"""CPU checks of the large-handle spot-check tooling (oracle/pyverify.py, orc_run_span): the span entry point of the oracle
is the streaming entry point fed the same bytes, and the sample of dongles covers the group / block edges."""
import numpy as np

import helpers
import pyoracle
import pyverify


def test_sample_covers_edges_and_is_deterministic():
    s = pyverify.sample_dongles(65536, 48)
    assert s == pyverify.sample_dongles(65536, 48) and len(s) == 48 and len(set(s)) == 48
    for d in (0, 15, 16, 127, 128, 129, 1023, 65535):
        assert d in s
    assert pyverify.sample_dongles(3, 10) == [0, 1, 2]
    assert pyverify.sample_dongles(200, 24)[-1] == 199


def test_run_span_is_run_device_on_the_same_bytes(pkg, built):
    for mixed, wave_rate in ((False, 8000), (True, 16000)):
        devices, carriers = helpers.plan_devices(1, mixed)
        nb = 3
        nbytes = helpers.stream_bytes(nb, wave_rate)
        iq = pkg.siggen.generate_u8(77, 0, nbytes // 2, carriers)
        a = pyoracle.Oracle(devices, wave_rate=wave_rate)
        b = pyoracle.Oracle(devices, wave_rate=wave_rate)
        ref = a.run_device(0, iq, nb)
        hop_bytes = 2 * round(2_560_000 / wave_rate)
        B = wave_rate // 8
        off = 0
        for k in range(nb):
            s = b.run_span(0, iq[off:])
            assert np.array_equal(s["waveout"].view(np.uint32), ref["waveout"][k].view(np.uint32))
            assert np.array_equal(s["trace"], ref["trace"][k]) and np.array_equal(s["axc"], ref["axc"][k])
            off += (B + 100) * hop_bytes if k == 0 else B * hop_bytes
        for j in range(8):
            assert a.stats(0, j) == b.stats(0, j)
