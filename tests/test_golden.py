"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py from the real reference).
CPU: the oracle must reproduce them bit for bit.  GPU (-m gpu): the HIP path must reproduce every squelch decision
and stay within 1e-4 RMS on the audio."""
import hashlib
import importlib
import json
import os
import sys

import numpy as np
import pytest

import helpers
import pyoracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import make_golden  # noqa: E402

CASES = sorted(make_golden.CASES)


def _fft_log(c):
    return c["format"][1] if "format" in c else 9


def _load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    c, devices, carriers, iq = make_golden.build_case(name)
    assert hashlib.sha256(iq.tobytes()).digest() == z["iq_sha256"].tobytes(), "synthetic I/Q generator no longer reproduces the fixture's input"
    assert json.loads(str(z["channels"])) == devices[0]["channels"]
    return z, c, devices, iq


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_golden(built, name):
    z, c, devices, iq = _load(name)
    orc = pyoracle.Oracle(devices, wave_rate=c["wave_rate"], fm_demod=c["fm_demod"], fft_log=_fft_log(c))
    got = orc.run_device(0, iq, c["n_batches"])
    assert got["n_batches"] == c["n_batches"]
    assert np.array_equal(got["axc"], z["axc"])
    assert np.array_equal(got["waveout"].view(np.uint32), z["waveout"].view(np.uint32))
    keep = z["iq_channels"]
    assert np.array_equal(got["iq_out"][:, keep].view(np.uint32), z["iq_out"].view(np.uint32))
    stats = json.loads(str(z["stats"]))
    for j, want in enumerate(stats):
        have = orc.stats(0, j)
        for k in ("open_count", "flappy_count", "ctcss_count", "no_ctcss_count", "active_counter", "bin"):
            assert have[k] == want[k], (j, k)
        for k in ("noise_level", "signal_level", "squelch_level", "agcavgfast"):
            assert np.float32(have[k]) == np.float32(want[k]), (j, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_golden(pkg, built, name):
    z, c, devices, iq = _load(name)
    with pkg.AirbandHip(devices, wave_rate=c["wave_rate"], fm_demod=c["fm_demod"], fft_log=_fft_log(c)) as hip:
        if "format" in c:
            assert hip.channelizer_name() == "dft_mfma_i8"  # every committed format case is one the matrix-core path claims
        iq = iq.view(np.uint8)
        pos = 0
        for b in range(c["n_batches"]):
            pos += hip.submit(0, iq[pos:])  # the staging ring holds ~5 batches, like the reference's input ring
            assert hip.process()
            out = hip.collect(iq=True, stats=True)
            assert np.array_equal(out["axc"], z["axc"][b]), "batch %d" % b
            assert helpers.rms(out["waveout"] - z["waveout"][b]) <= 1e-4
            keep = z["iq_channels"]
            if len(keep):
                assert helpers.rms(out["iq_out"][keep] - z["iq_out"][b]) <= 1e-4 * max(1.0, helpers.rms(z["iq_out"][b]))
        stats = json.loads(str(z["stats"]))
        for j, want in enumerate(stats):
            for k in ("open_count", "flappy_count", "ctcss_count", "no_ctcss_count", "active_counter", "bin"):
                assert out["stats"][j][k] == want[k], (j, k)
            assert abs(out["stats"][j]["noise_level"] - want["noise_level"]) <= 1e-4 * want["noise_level"]
