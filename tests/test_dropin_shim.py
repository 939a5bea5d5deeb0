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


INPUT_DISABLED = 5  # input_state_t (src/input-common.h:34)


def _failure_run(pkg, hip_lib):
    n_dev, n_batches, wave_rate = 3, 8, 16000
    devices, carriers = helpers.plan_devices(n_dev, True, None)
    nbytes = helpers.stream_bytes(n_batches, wave_rate) + 4 * 640
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    return pyref.run_reference_all(devices, iq, n_batches, nfm=True, hip_lib=hip_lib, fail_after=[None, 3, None], end_of_streams=True)


@pytest.mark.skipif(not (pyref.have_ref(True) and os.path.exists(pyref.ref_lib_path(True, "patched"))), reason="oracle/_ref not built")
def test_one_input_at_end_of_file_then_all(pkg, built):
    """What demodulate() does when inputs stop (src/rtl_airband.cpp:377-391; the file input sets INPUT_FAILED at end of file,
    src/input-file.cpp:101-111): device 1 of 3 fails after 3 batches -> its outputs are disabled, devices_running drops to 2, the
    other two devices keep producing exactly what the reference produces; when the last input ends the demodulator sets do_exit and
    returns.  Run once with demodulate(), once with demodulate_hip() from the patched reference."""
    ref = _failure_run(pkg, None)
    hip = _failure_run(pkg, pkg.LIB_PATH)
    for r in (ref, hip):
        assert r["batches"] == [8, 3, 8]
        assert r["outputs_disabled"] == [0, 1, 0] and r["devices_running"] == 2 and r["input_state"][1] == INPUT_DISABLED
        assert r["exited_on_its_own"], "the demodulator kept spinning after every input had failed"
        assert r["devices_running_at_exit"] == 0 and r["outputs_disabled_at_exit"] == [1, 1, 1]
    for d, nb in enumerate(ref["batches"]):
        assert np.array_equal(ref["axc"][d, :nb], hip["axc"][d, :nb]), d
        assert helpers.rms(ref["waveout"][d, :nb] - hip["waveout"][d, :nb]) <= 1e-4
        for j in range(8):
            a, b = ref["stats"][d][j], hip["stats"][d][j]
            for k in ("open_count", "flappy_count", "ctcss_count", "no_ctcss_count", "active_counter"):
                assert a[k] == b[k], (d, j, k, a[k], b[k])
    assert (ref["axc"][0] == ord("*")).any() and (ref["axc"][2] == ord("*")).any()


@pytest.mark.skipif(not (pyref.have_ref(True) and os.path.exists(pyref.ref_lib_path(True, "patched"))), reason="oracle/_ref not built")
def test_shard_with_two_device_classes(pkg, built):
    """The reference takes sample rate and sample format per device (src/rtl_airband.cpp:394,402-455): one demodulate() shard with a
    2.56 MS/s u8 dongle and a 2.4 MS/s CS16 (SoapySDR) device.  The shim runs one library handle per class."""
    capi = pkg.capi
    n_batches, wave_rate = 6, 16000
    d0, iq0 = helpers.format_case(pkg, capi.SFMT_U8, 9, 2_560_000, wave_rate, 1, n_batches)
    d1, iq1 = helpers.format_case(pkg, capi.SFMT_S16, 9, 2_400_000, wave_rate, 1, n_batches, first_dongle=1)
    devices, iq = d0 + d1, [iq0[0], iq1[0].view(np.uint8)]
    ref = pyref.run_reference_all(devices, iq, n_batches, nfm=True)
    hip = pyref.run_reference_all(devices, iq, n_batches, nfm=True, hip_lib=pkg.LIB_PATH)
    assert ref["batches"] == hip["batches"] == [n_batches, n_batches]
    assert np.array_equal(ref["axc"], hip["axc"])
    for d in range(2):
        assert (ref["axc"][d] == ord("*")).any()
        assert helpers.rms(ref["waveout"][d] - hip["waveout"][d]) <= 1e-4
        for j in range(8):
            a, b = ref["stats"][d][j], hip["stats"][d][j]
            for k in ("open_count", "flappy_count", "ctcss_count", "no_ctcss_count", "active_counter", "bin"):
                assert a[k] == b[k], (d, j, k, a[k], b[k])
