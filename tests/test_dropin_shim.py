"""The drop-in claim, end to end: the REAL reference harness (its own device_t / channel_t / input_t objects, circbuffer_append
on the producer side, the output thread's waveavail protocol on the consumer side) run twice -- once with the reference's
demodulate(), once built from the PATCHED reference: integration/airband_hip.patch applied to a scratch copy of the reference
sources, integration/demod_hip.cpp (the translation unit the patch adds) compiled verbatim, demodulate_hip() started instead of
demodulate().  Statistics of the second run are read through the reference's own Squelch getters, i.e. the way the stats file and
the TUI read them (src/output.cpp:617-761), fed by the patch's Squelch::mirror().  Needs the prebuilt oracle/_ref (it travels to
the GPU box) and a GPU."""
import os
import numpy as np
import pytest

import helpers
import pyref

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not (pyref.have_ref(True) and os.path.exists(pyref.ref_lib_path(True, "patched"))), reason="oracle/_ref not built")
@pytest.mark.parametrize("mixed,wave_rate", [(False, 8000), (True, 16000)])
def test_reference_harness_with_hip_backend(pkg, built, mixed, wave_rate):
    def tweak(d, ch):
        if mixed:
            ch[3]["has_iq_outputs"] = 1
            ch[0]["bandwidth_hz"] = 8000
    n_dev, n_batches = 3, 12
    devices, carriers = helpers.plan_devices(n_dev, mixed, tweak)
    nbytes = helpers.stream_bytes(n_batches, wave_rate) + 4 * 640  # a little slack: demodulate() wants one extra hop queued
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    nfm = wave_rate == 16000
    if not nfm:
        pytest.importorskip("numpy")
    ref = pyref.run_reference_all(devices, iq, n_batches, nfm=nfm)
    hip = pyref.run_reference_all(devices, iq, n_batches, nfm=nfm, hip_lib=pkg.LIB_PATH)
    assert ref["n_batches"] == n_batches and hip["n_batches"] == n_batches
    assert np.array_equal(ref["axc"], hip["axc"]), "squelch decisions differ between demodulate() and the HIP backend"
    assert (ref["axc"] == ord("*")).any()
    assert helpers.rms(ref["waveout"] - hip["waveout"]) <= 1e-4
    assert helpers.rms(ref["iq_out"] - hip["iq_out"]) <= 1e-4 * max(1.0, helpers.rms(ref["iq_out"]))
    for d in range(n_dev):
        for j in range(8):
            a, b = ref["stats"][d][j], hip["stats"][d][j]
            for k in ("open_count", "flappy_count", "ctcss_count", "no_ctcss_count", "active_counter", "bin"):
                assert a[k] == b[k], (d, j, k, a[k], b[k])
            for k in ("noise_level", "signal_level", "squelch_level", "agcavgfast"):  # through Squelch::mirror() and the getters
                assert abs(a[k] - b[k]) <= 1e-4 * max(abs(a[k]), 1e-2), (d, j, k, a[k], b[k])
